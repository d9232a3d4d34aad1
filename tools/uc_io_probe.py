#!/usr/bin/env python3
"""Does it help K1 / the chain if the streamed buffers (input ring, chain outputs) bypass the caches?  The workspaces of the
launch groups in flight (207 MB) live in the 256 MB Infinity Cache; inputs and outputs stream through it once.  This
probe runs bench.py with every buffer the HOST side allocates (ring, outputs - not the library's workspaces) taken from
hipExtMallocWithFlags(flag): 0 = default, 1 = fine-grained, 3 = uncached.
python tools/uc_io_probe.py <flag> [bench.py arguments]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "impulcifer-pip313_amd"))
from impulse_hip import _native  # noqa: E402

flag = int(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
if flag:
    hip = C.CDLL("libamdhip64.so")
    special = set()
    plain_malloc, plain_free = _native.Context.malloc, _native.Context.free

    def malloc(self, nbytes):
        if nbytes < (1 << 20):
            return plain_malloc(self, nbytes)
        p = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(int(nbytes)), C.c_uint(flag))
        if rc != 0:
            raise RuntimeError(f"hipExtMallocWithFlags({nbytes}, {flag}) -> {rc}")
        special.add(p.value)
        return p.value

    def free(self, p):
        if int(p) in special:
            self.synchronize()
            hip.hipFree(C.c_void_p(int(p)))
            special.discard(int(p))
        else:
            plain_free(self, p)

    _native.Context.malloc, _native.Context.free = malloc, free
import bench  # noqa: E402

sys.exit(bench.main())
