import os, sys, ctypes, time, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from gkmqc_amd import device
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
a = bench.parse_args([])
tmp = tempfile.mkdtemp(); pf, nf = bench.write_problem_files(a, a.n_pos, a.n_neg, tmp)
n = a.n_pos + a.n_neg
kmat = np.zeros((n, n)); rows = (kmat.ctypes.data + np.arange(n) * kmat.strides[0]).astype(np.uintp); sizes = np.zeros(2, dtype=np.int32)
opt = device.gkmOpt(4, 11, 7, 3, 50, 50.0, 1.0, os.fsencode(pf), os.fsencode(nf), 16, 3)
lib = device.load()
for i in range(2):
    t0 = time.perf_counter(); rc = lib.gkm_main_pywrapper(ctypes.byref(opt), rows.ctypes.data, sizes.ctypes.data); print("call %d: %.1f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
