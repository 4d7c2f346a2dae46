import sys, os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, qchem_rs_amd as q
from conftest import load_system
mol,basis=sys.argv[1],sys.argv[2]
m=load_system(mol,basis); s=q.System(m)
st=q.ScfStepper(s)
for it in range(25):
    e,rms=st.iterate()
    print('ITER',it,'%.12f'%e,'%.3e'%rms, file=sys.stderr)
    if rms<1e-11: break
