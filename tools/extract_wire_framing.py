#!/usr/bin/env python3
"""Prints the framing of the reference's key fixtures (everything except the key vectors themselves:
tfhe-rs safe_serialize header, version tags, parameter structs) as the base64 constants that
fhe_amd/tfhers_wire.py embeds.  Input: tests/golden/privateKey.bin, publicKey.bin (copies of
sdk/relayer/src/test/keys/ of the reference)."""
import base64
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fhe_amd import tfhers_wire as w  # noqa: E402


def pieces(blob, vecs):
    out, pos = [], 0
    for off, v in vecs:
        out.append(blob[pos:off])
        pos = off + 8 * len(v)
    out.append(blob[pos:])
    return out


priv = open(os.path.join(ROOT, "tests", "golden", "privateKey.bin"), "rb").read()
pub = open(os.path.join(ROOT, "tests", "golden", "publicKey.bin"), "rb").read()
pv = w._u64_vectors(priv, lambda v: int(v.max(initial=0)) <= 1 and 0 < int(v.sum()) < len(v))
qv = w._u64_vectors(pub, lambda v: len(v) % 2 == 0 and int((v >> np.uint64(48)).astype(bool).mean() > 0.9), min_len=1024)
print("CLIENT_KEY_FRAMING = (")
for p in pieces(priv, pv):
    print(f'    "{base64.b64encode(p).decode()}",')
print(")\nCOMPACT_PUBLIC_KEY_FRAMING = (")
for p in pieces(pub, qv):
    print(f'    "{base64.b64encode(p).decode()}",')
print(")")
