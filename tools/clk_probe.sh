python - <<'PY' &
import sys, torch, time
sys.path.insert(0, '.')
from multimodaltopicsegmentation_amd import _lib as L, ops
dev='cuda'
M,N,K=8192,7168,8192
A=torch.randn(M,K,device=dev).to(torch.bfloat16); B=torch.randn(N,K,device=dev).to(torch.bfloat16); out=torch.empty(M,N,dtype=torch.bfloat16,device=dev)
for name, fn in (('ours', lambda: ops.gemm(L.NT,A,B,out,M=M,N=N,K=K)), ('vendor', lambda: torch.matmul(A,B.t()))):
    t0=time.time()
    while time.time()-t0 < 6:
        for _ in range(50): fn()
        torch.cuda.synchronize()
    print('done', name, flush=True)
PY
PID=$!
sleep 2.5
for i in 1 2 3 4 5 6 7 8 9 10; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | tr '\n' ' '; echo; sleep 1; done
wait $PID
