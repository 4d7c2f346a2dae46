#!/usr/bin/env python3
"""Author cc-pVDZ / cc-pVTZ (H, C, O) in MolSSI-BSE JSON schema 0.1 and the O2 molecule file.

BASELINE.json's configs name cc-pVDZ/cc-pVTZ and O2, none of which the reference ships and there
is no network.  The numbers below are Dunning's published correlation-consistent sets (T.H. Dunning Jr.,
J. Chem. Phys. 90, 1007 (1989)) entered by hand in the *full general-contraction* form.  They are validated in
tests/test_oracle_known_answers.py by a literature total energy (H2O RHF/cc-pVDZ).  GPU-vs-oracle parity does not
depend on the provenance of these numbers - both paths read the same file.
"""
import json, os
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "data")

def unit(n, i):
    return [1.0 if k == i else 0.0 for k in range(n)]

def shell(L, exps, rows, ftype):
    return {"function_type": ftype, "region": "", "angular_momentum": [L],
            "exponents": ["%.10E" % e for e in exps],
            "coefficients": [["%.10E" % c for c in r] for r in rows]}

def S(exps, rows): return shell(0, exps, rows, "gto")
def P(exps, rows): return shell(1, exps, rows, "gto")
def D(e): return shell(2, [e], [[1.0]], "gto_spherical")
def F(e): return shell(3, [e], [[1.0]], "gto_spherical")

ccpvdz = {
 "1": [S([13.01, 1.962, 0.4446, 0.122],
         [[0.019685, 0.137977, 0.478148, 0.501240], unit(4, 3)]),
       P([0.727], [[1.0]])],
 "6": [S([6665.0, 1000.0, 228.0, 64.71, 21.06, 7.495, 2.797, 0.5215, 0.1596],
         [[0.000692, 0.005329, 0.027077, 0.101718, 0.274740, 0.448564, 0.285074, 0.015204, -0.003191],
          [-0.000146, -0.001154, -0.005725, -0.023312, -0.063955, -0.149981, -0.127262, 0.544529, 0.580496],
          unit(9, 8)]),
       P([9.439, 2.002, 0.5456, 0.1517],
         [[0.038109, 0.209480, 0.508557, 0.468842], unit(4, 3)]),
       D(0.55)],
 "8": [S([11720.0, 1759.0, 400.8, 113.7, 37.03, 13.27, 5.025, 1.013, 0.3023],
         [[0.000710, 0.005470, 0.027837, 0.104800, 0.283062, 0.448719, 0.270952, 0.015458, -0.002585],
          [-0.000160, -0.001263, -0.006267, -0.025716, -0.070924, -0.165411, -0.116955, 0.557368, 0.572759],
          unit(9, 8)]),
       P([17.70, 3.854, 1.046, 0.2753],
         [[0.043018, 0.228913, 0.508728, 0.460531], unit(4, 3)]),
       D(1.185)],
}
ccpvtz = {
 "1": [S([33.87, 5.095, 1.159, 0.3258, 0.1027],
         [[0.006068, 0.045308, 0.202822, 0.503903, 0.383421], unit(5, 3), unit(5, 4)]),
       P([1.407, 0.388], [unit(2, 0), unit(2, 1)]),
       D(1.057)],
 "6": [S([8236.0, 1235.0, 280.8, 79.27, 25.59, 8.997, 3.319, 0.9059, 0.3643, 0.1285],
         [[0.000531, 0.004108, 0.021087, 0.081853, 0.234817, 0.434401, 0.346129, 0.039378, -0.008983, 0.002385],
          [-0.000113, -0.000878, -0.004540, -0.018133, -0.055760, -0.126895, -0.170352, 0.140382, 0.598684, 0.395389],
          unit(10, 7), unit(10, 9)]),
       P([18.71, 4.133, 1.200, 0.3827, 0.1209],
         [[0.014031, 0.086866, 0.290216, 0.501008, 0.343406], unit(5, 3), unit(5, 4)]),
       D(1.097), D(0.318), F(0.761)],
 "8": [S([15330.0, 2299.0, 522.4, 147.3, 47.55, 16.76, 6.207, 1.752, 0.6882, 0.2384],
         [[0.000508, 0.003929, 0.020243, 0.079181, 0.230687, 0.433118, 0.350260, 0.042728, -0.008154, 0.002381],
          [-0.000115, -0.000895, -0.004636, -0.018724, -0.058463, -0.136463, -0.175740, 0.160934, 0.603418, 0.378765],
          unit(10, 7), unit(10, 9)]),
       P([34.46, 7.749, 2.280, 0.7156, 0.2140],
         [[0.015928, 0.099740, 0.310492, 0.491026, 0.336337], unit(5, 3), unit(5, 4)]),
       D(2.314), D(0.645), F(1.428)],
}

def write(name, table):
    doc = {"molssi_bse_schema": {"schema_type": "complete", "schema_version": "0.1"},
           "name": name, "description": name + " (H, C, O only; authored offline, see tools/author_ccpvxz.py)",
           "function_types": ["gto", "gto_spherical"],
           "_provenance": "hand-entered Dunning 1989 sets, full general contraction; NOT a reference data file",
           "elements": {z: {"electron_shells": sh} for z, sh in table.items()}}
    with open(os.path.join(OUT, "basis", name + ".json"), "w") as f:
        json.dump(doc, f, indent=1)

write("cc-pVDZ", ccpvdz)
write("cc-pVTZ", ccpvtz)

ANG = 1.0 / 0.52917721067   # bohr per angstrom (CODATA 2014, the value PySCF uses)
def mol(name, atoms):
    with open(os.path.join(OUT, "mol", name + ".json"), "w") as f:
        json.dump([{"element": str(z), "position": list(p)} for z, p in atoms], f, indent=4)

# O2 at r = 1.2075 A on the z axis; bohr like every reference molecule file
mol("oxygen", [(8, (0.0, 0.0, 1.1409)), (8, (0.0, 0.0, -1.1409))])
# near-equilibrium water used by many quantum-chemistry tutorials (angstrom -> bohr); literature-energy check only
mol("water_eq", [(8, (0.0, 0.0, 0.0)), (1, (0.0, -0.757 * ANG, 0.587 * ANG)), (1, (0.0, 0.757 * ANG, 0.587 * ANG))])
# T.D. Crawford's programming-project water geometry (bohr), STO-3G SCF energy published to 12 digits
mol("water_crawford", [(8, (0.0, -0.143225816552, 0.0)), (1, (1.638036840407, 1.136548822547, 0.0)),
                       (1, (-1.638036840407, 1.136548822547, 0.0))])
print("ok")
