import sys, time, numpy as np
sys.path.insert(0, '/root/repo' if len(sys.argv) < 2 else sys.argv[1])
from mfrec_amd import _shim as S, synth
from mfrec_amd.lib.datasets import balanced_order_gpu, balanced_order
import ctypes as C
S.init(0)
idx, r = synth.make_ratings(100000, 50000, 10000000)
for rep in range(2):
    t0 = time.time(); perm, rounds = balanced_order_gpu(idx, 100000, 50000); dt = time.time() - t0
    print("balanced_order_gpu: %.1f ms, rounds %d" % (dt * 1e3, rounds))
lev = np.zeros(len(idx), dtype=np.int32); n = C.c_int64(0)
b = np.ascontiguousarray(idx[perm])
S.check(S.lib().mfx_schedule_levels(S.i32(b, "idx", 2), len(b), 100000, 50000, lev.ctypes.data_as(S._ip), C.byref(n)))
print("levels of the balanced order:", n.value, "max degree", max(np.bincount(idx[:,0]).max(), np.bincount(idx[:,1]).max()))
