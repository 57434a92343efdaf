import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as L, ops
dev='cuda'
g=torch.Generator(device=dev).manual_seed(0)
K=16384
def t(fn):
    best=1e9
    for _ in range(3):
        fn(); fn(); torch.cuda.synchronize()
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        e.record(); torch.cuda.synchronize()
        best=min(best,s.elapsed_time(e)*1e3/20)
    return best
for M,N in ((256,1792),(1792,256)):
    dy=torch.randn(K,M,device=dev,generator=g).to(torch.bfloat16)
    x=torch.randn(K,N,device=dev,generator=g).to(torch.bfloat16)
    out=torch.empty(M,N,device=dev)
    for tile in (0,128,224,256):
        for sp in (0,16,24,32):
            L.check(L.lib.mts_set_option(b'gemm_tile',tile)); L.check(L.lib.mts_set_option(b'gemm_splits',sp))
            try:
                us=t(lambda: ops.linear_wgrad(dy,x,out))
                import ctypes
                a,b=ctypes.c_int(0),ctypes.c_int(0); L.lib.mts_gemm_last_plan(ctypes.byref(a),ctypes.byref(b))
                print(f'M={M} N={N} tile_opt={tile} splits_opt={sp}: {us:6.1f} us  (ran tile {a.value} splits {b.value})',flush=True)
            except Exception as ex:
                print('fail',tile,sp,str(ex)[:80])
L.lib.mts_set_option(b'gemm_tile',0); L.lib.mts_set_option(b'gemm_splits',0)
