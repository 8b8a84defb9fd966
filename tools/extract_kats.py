#!/usr/bin/env python3
"""Extracts the operator known-answer vectors from the reference's test suite as data.

Reads tests/fhevm-suite/e2e/test/fhevmOperations/fhevmOperations{1..13}.ts as TEXT (nothing is
executed) and writes tests/golden/fhevm_operator_kats.json: one record per test with the
operator, operand types, clear inputs and the decrypted result the reference expects.
Usage: python tools/extract_kats.py [/root/reference]
"""
import glob
import json
import os
import re
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
src_dir = os.path.join(ref, "tests/fhevm-suite/e2e/test/fhevmOperations")
out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/fhevm_operator_kats.json")

title = re.compile(r"""it\('test operator "(\w+)" overload \(([^)]*)\) => (\w+) test (\d+) \(([^)]*)\)'""")
expect = re.compile(r"expect\(res\)\.to\.equal\((\w+)\)")

records = []
for path in sorted(glob.glob(os.path.join(src_dir, "fhevmOperations*.ts"))):
    name = os.path.basename(path)
    lines = open(path).read().split("\n")
    cur = None
    for ln, line in enumerate(lines, 1):
        m = title.search(line)
        if m:
            cur = {"op": m.group(1), "types": [t.strip() for t in m.group(2).split(",")], "result_type": m.group(3),
                   "inputs": [s.strip() for s in m.group(5).split(",")], "src": f"{name}:{ln}"}
            continue
        if cur is not None:
            e = expect.search(line)
            if e:
                v = e.group(1)
                cur["expected"] = v[:-1] if v.endswith("n") else v
                records.append(cur)
                cur = None
with open(out_path, "w") as f:
    json.dump({"source": "luxfi/fhe tests/fhevm-suite/e2e/test/fhevmOperations/*.ts (decrypted results)",
               "count": len(records), "kats": records}, f, indent=0)
print(len(records), "vectors ->", out_path)
