#!/usr/bin/env python3
"""Generates tests/golden/abi_reference.json: struct layouts and parameter unpacking as the REFERENCE's own header gives them
(include/NativeUtils/depthprocessing.h:29-48,50-98, include/NativeUtils/icp.h:15-18,65), printed by oracle/_ref/ref_abi, which
oracle/Makefile compiles from oracle/ref_abi_harness.cpp against the header where it lies under /root/reference (that build also
static_asserts this repository's prototypes against the reference's).  Run in the build container (needs /root/reference):

    python tests/golden/make_abi_golden.py

The fixture is data only (numbers the reference's compiler and constructors produced)."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
out = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "ref_abi")], text=True)
data = json.loads(out)
data["generator"] = "tests/golden/make_abi_golden.py (oracle/_ref/ref_abi = the reference's header compiled in place)"
with open(os.path.join(ROOT, "tests", "golden", "abi_reference.json"), "w") as f:
    json.dump(data, f, indent=1)
    f.write("\n")
print(json.dumps(data, indent=1))
