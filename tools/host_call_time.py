"""Host time of single C-ABI calls (ctypes marshalling + the launches inside), GPU kept idle-ish by tiny tensors: python tools/host_call_time.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccst_amd import _lib, ops
from ccst_amd._lib import check, ptr, stream_ptr
dev = torch.device("cuda:0")
lib = _lib.load()
M, C = 64 * 14 * 14, 256
x = torch.randn(M, C, device=dev)
y = torch.empty_like(x)
g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
save = torch.empty(2, C, device=dev)
ws = torch.empty(int(lib.ccst_bn_workspace_bytes(M, C)), device=dev, dtype=torch.uint8)
words = torch.zeros(64, device=dev, dtype=torch.int32)
stats = torch.zeros(64, C, 2, device=dev)


def timeit(name, fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-44s host %6.1f us/call   (until done %6.1f us/call)" % (name, (t1 - t0) * 1e6 / n, (t2 - t0) * 1e6 / n))


timeit("ccst_fill_f32 (1 launch)", lambda: lib.ccst_fill_f32(ptr(y), 0.0, 1024, stream_ptr()))
timeit("bn_train_fwd_mask, stats given (2 launches)", lambda: lib.ccst_bn_train_fwd_mask_f32(ptr(x), ptr(g), ptr(b), ptr(rm), ptr(rv), 0.1, 1e-5, None, 1, ptr(y), None,
                                                                                  ptr(save[0]), ptr(save[1]), M, C, ptr(stats), 64, ptr(ws), ws.numel(), ptr(words), stream_ptr()))
timeit("bn_train_fwd_mask, no stats (3 launches)", lambda: lib.ccst_bn_train_fwd_mask_f32(ptr(x), ptr(g), ptr(b), ptr(rm), ptr(rv), 0.1, 1e-5, None, 1, ptr(y), None,
                                                                               ptr(save[0]), ptr(save[1]), M, C, None, 0, ptr(ws), ws.numel(), ptr(words), stream_ptr()))
timeit("torch.empty_like", lambda: torch.empty_like(x))
timeit("ops.absmax_words", lambda: ops.absmax_words(dev))
timeit("torch.cuda.Event().record()", lambda: torch.cuda.Event().record())
s2 = torch.cuda.Stream()


def switch():
    with torch.cuda.stream(s2):
        pass


timeit("with torch.cuda.stream(side): pass", switch)
timeit("side.wait_stream(main)", lambda: s2.wait_stream(torch.cuda.current_stream()))
timeit("x.record_stream(side)", lambda: x.record_stream(s2))
