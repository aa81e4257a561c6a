"""Where the multi-tile projection kernel's time goes (config-4 shard): the kernel as it runs, with the X operand from cache
(MFMA + L2 operand alone) and without the MFMAs (the loads alone).  python tools/probe_l2w.py"""
import ctypes, os, sys, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from animsnapbases_amd import posComponents, posSnapshots

N, F, K = 100000, 2000, 128
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(1)
Xd = torch.rand((F, N, 3), dtype=torch.float64, device=dev, generator=gen) * 2 - 1
snaps = posSnapshots.from_device(Xd.data_ptr(), F, N, rest_shape="first", standarize=True)
param = types.SimpleNamespace(vertPos_bases_type="PCA", q_standarize=True, q_massWeight=False, q_orthogonal=False, q_support="global",
                              vertPos_numComponents=K, store_vertPos_PCA_sing_val=False, vertPos_smooth_min_dist=0.1,
                              vertPos_smooth_max_dist=0.25, vertPos_rest_shape="first", name="p", vertPos_output_directory=".")
comp = posComponents(param, snaps)
comp.deflate_mode = "project"
comp.extract_k_components(None)           # projection mode set up, component storage allocated
eng = snaps._engine
for nct in (2, 3, 4):
    out = []
    for mode in (0, 1, 2):
        ms = ctypes.c_double()
        eng._ck(eng.lib.asb_test_l2w_probe(eng.h, nct, mode, 5, ctypes.byref(ms)))
        out.append(ms.value)
    print("tiles %d: as it runs %.3f ms | X from cache %.3f ms | loads alone %.3f ms" % (nct, *out), flush=True)
out = []
for mode in (10, 13, 14):
    ms = ctypes.c_double()
    eng._ck(eng.lib.asb_test_l2w_probe(eng.h, 4, mode, 5, ctypes.byref(ms)))
    out.append(ms.value)
print("k_project_l2c<4,3>: as it runs %.3f ms | no X loads inside the loop %.3f ms | nor LDS reads of the weights %.3f ms" % tuple(out), flush=True)
out = []
for mode in (110, 113, 114):
    ms = ctypes.c_double()
    eng._ck(eng.lib.asb_test_l2w_probe(eng.h, 4, mode, 5, ctypes.byref(ms)))
    out.append(ms.value)
print("the same with pseudo-random weights instead of constants: %.3f | %.3f | %.3f ms" % tuple(out), flush=True)
for nct in (2, 3):
    ms = ctypes.c_double()
    eng._ck(eng.lib.asb_test_l2w_probe(eng.h, nct, 100, 5, ctypes.byref(ms)))
    print("k_project_l2w, %d tiles, pseudo-random weights: as it runs %.3f ms" % (nct, ms.value), flush=True)
ms = ctypes.c_double()
eng._ck(eng.lib.asb_test_l2w_probe(eng.h, 4, 15, 5, ctypes.byref(ms)))
print("k_project_l2c<4,3> without the barrier per stage (wrong results, timing only): %.3f ms" % ms.value, flush=True)
out = []
for mode in (20, 23, 24, 25):
    ms = ctypes.c_double()
    eng._ck(eng.lib.asb_test_l2w_probe(eng.h, 4, mode, 5, ctypes.byref(ms)))
    out.append(ms.value)
print("k_project_l2d<4,3> (round 3): as it runs %.3f ms | no X loads inside the loop %.3f ms | nor LDS reads of the weights %.3f ms | "
      "as it runs without the stage synchronisation (wrong results, timing only) %.3f ms" % tuple(out), flush=True)
