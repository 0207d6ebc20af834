"""Writes tests/golden/uff_headers.json: for every header of tests/uff_cases.py, the answer implied by the reference's
XmlDescriptor (oracle/_ref: xml_descriptor.cpp compiled unmodified against Qt 5.9.7 QtXml).  Run here; the fixture travels."""
import ctypes as C
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle_lib as ol  # noqa: E402
import test_iqfile_probe as tp  # noqa: E402
import pathlib  # noqa: E402

R = ol.ref()
out = {}
with tempfile.TemporaryDirectory() as d:
    for name, path, flen in tp._uff_files(pathlib.Path(d)):
        if name.startswith("empty") or name.startswith("not_xml"):
            out[name] = None
            continue
        ints = np.zeros(5, np.int32)
        strs = C.create_string_buffer(48)
        nel = C.c_longlong(0)
        assert R.ref_uff_describe(path.encode(), ints, strs, C.byref(nel)) == 0
        s3 = [strs.raw[i:i + 16].split(b"\0")[0].decode() for i in (0, 16, 32)]
        w = tp._uff_expect_from_descriptor(ints.tolist(), s3, nel.value, flen)
        out[name] = list(w) if w else None
json.dump(out, open(os.path.join(HERE, "uff_headers.json"), "w"), indent=0, sort_keys=True)
print("wrote uff_headers.json:", len(out), "cases,", sum(v is not None for v in out.values()), "accepted")
