#!/usr/bin/env python3
"""The toolchain the bit-exactness evidence was produced on, as an executable pin.

Thirteen float -> float sites of the device's fp64 trig calls are identical to the host's glibc "by exhaustion" (tools/exhaustive_trig.py,
profiles/r02/exhaustive_trig.json: all 2^32 inputs of each, 0 differences) -- a statement about ONE device math library (ROCm's ocml
bitcode, inlined by hipcc) and ONE host libm (glibc's libm.so.6).  A different hipcc / ocml / glibc silently voids it.  This module
names what those are:

    current()   what this machine has: hipcc --version, sha256 of ocml.bc, the glibc release, sha256 of libm.so.6
    PIN         profiles/toolchain_pin.json: what the exhaustive sweeps and the golden vectors were produced on
    __graft_entry__.build() writes current() next to the library (dj_brdf_amd/lib/toolchain.json);
    tests/test_toolchain_pin.py FAILS when either differs from the pin.

Re-pin after a toolchain bump (one GPU session, ~25 min):
    gpurun --timeout 2400 -- 'python tools/exhaustive_trig.py'  &&  python tools/toolchain_pin.py --accept
"""
import ctypes.util
import hashlib
import json
import os
import platform
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN = os.path.join(ROOT, "profiles", "toolchain_pin.json")
BUILT = os.path.join(ROOT, "dj_brdf_amd", "lib", "toolchain.json")
KEYS = ("hipcc", "ocml_sha256", "glibc", "libm_sha256")


def _sha(path):
    try:
        h = hashlib.sha256()
        with open(path, "rb") as f:
            for blk in iter(lambda: f.read(1 << 20), b""):
                h.update(blk)
        return h.hexdigest()
    except OSError:
        return None


def _libm_path():
    for p in ("/lib/x86_64-linux-gnu/libm.so.6", "/lib64/libm.so.6", "/usr/lib/x86_64-linux-gnu/libm.so.6"):
        if os.path.exists(p):
            return os.path.realpath(p)
    return None


def current():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    try:
        out = subprocess.run([hipcc, "--version"], capture_output=True, text=True, timeout=60).stdout.splitlines()
        ver = " | ".join(l.strip() for l in out if l.startswith("HIP version") or "clang version" in l)
    except Exception:
        ver = None
    rocm = os.path.realpath(os.path.join(os.path.dirname(os.path.realpath(hipcc)), ".."))
    ocml = None
    for cand in (os.path.join(rocm, "amdgcn", "bitcode", "ocml.bc"), "/opt/rocm/amdgcn/bitcode/ocml.bc"):
        if os.path.exists(cand):
            ocml = cand
            break
    try:
        glibc = os.confstr("CS_GNU_LIBC_VERSION")
    except (ValueError, OSError):
        glibc = " ".join(platform.libc_ver())
    return {"hipcc": ver, "ocml_sha256": _sha(ocml) if ocml else None, "glibc": glibc, "libm_sha256": _sha(_libm_path()) if _libm_path() else None}


def differences(have, want):
    return ["%s: %r, the evidence was produced on %r" % (k, have.get(k), want.get(k)) for k in KEYS if have.get(k) != want.get(k)]


HOWTO = ("re-run the exhaustive sweeps on the new toolchain and re-pin:  gpurun --timeout 2400 -- 'python tools/exhaustive_trig.py'  "
         "&&  python tools/toolchain_pin.py --accept   (INTEGRATION.md section 5)")

if __name__ == "__main__":
    cur = current()
    if "--accept" in sys.argv:
        cur["note"] = ("the toolchain tools/exhaustive_trig.py (profiles/r02/exhaustive_trig.json), the libm restatement tests and "
                       "tests/golden/*.npz were produced on; tests/test_toolchain_pin.py fails when the machine or the built library differs")
        json.dump(cur, open(PIN, "w"), indent=1)
        print("pinned", PIN)
    else:
        print(json.dumps(cur, indent=1))
        if os.path.exists(PIN):
            d = differences(cur, json.load(open(PIN)))
            print("matches the pin" if not d else "DIFFERS from the pin:\n  " + "\n  ".join(d) + "\n" + HOWTO)
            sys.exit(1 if d else 0)
