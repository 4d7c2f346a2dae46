#!/usr/bin/env python3
"""Copy the reference's *data* files this build consumes into data/ (inputs only, no code).

The reference (MIT, /root/reference/LICENSE) ships MolSSI-BSE JSON basis sets and toy molecule JSON files
(SURVEY.md App. C).  /root/reference does not exist on the GPU box, so the handful of inputs the parity tests and
bench.py need travel with this repo.  Basis files are trimmed to the elements that occur in the shipped molecules
(H, C, O, Cl) - the per-element records are copied verbatim (strings untouched), everything else in the BSE schema
that the loader reads (`molssi_bse_schema`, `name`, `function_types`) is kept.

Run here (container with /root/reference mounted):  python tools/make_fixtures.py
"""
import json, os, shutil, sys
REF = "/root/reference/data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "data")
KEEP_ELEMENTS = ["1", "6", "8", "17"]
BASES = ["STO-3G", "6-31G", "6-31G_st_st", "6-311++G_st_st", "def2-SV(P)"]

def main():
    os.makedirs(os.path.join(OUT, "basis"), exist_ok=True)
    os.makedirs(os.path.join(OUT, "mol"), exist_ok=True)
    for b in BASES:
        src = json.load(open(os.path.join(REF, "basis", b + ".json")))
        dst = {k: src[k] for k in ("molssi_bse_schema", "name", "description", "function_types", "family") if k in src}
        dst["elements"] = {}
        for z in KEEP_ELEMENTS:
            if z in src["elements"]:
                e = src["elements"][z]
                dst["elements"][z] = {"electron_shells": e["electron_shells"]}
        dst["_provenance"] = ("trimmed copy (elements %s) of the reference data file data/basis/%s.json; "
                              "element records verbatim" % (",".join(KEEP_ELEMENTS), b))
        with open(os.path.join(OUT, "basis", b + ".json"), "w") as f:
            json.dump(dst, f, indent=1)
    for m in os.listdir(os.path.join(REF, "mol")):
        shutil.copyfile(os.path.join(REF, "mol", m), os.path.join(OUT, "mol", m))
        os.chmod(os.path.join(OUT, "mol", m), 0o644)
    print("fixtures written to", os.path.normpath(OUT))

if __name__ == "__main__":
    sys.exit(main())
