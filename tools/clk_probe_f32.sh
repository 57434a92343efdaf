# Shader clock under the fp32 (parity mode) MFMA GEMM, next to the bf16 one: rocm-smi once a second while each kernel loops for 5 s.
python - <<'PY' &
import sys, torch, time
sys.path.insert(0, '.')
from multimodaltopicsegmentation_amd import _lib as L, ops
dev='cuda'
M,N,K=16384,5376,1792
A=torch.randn(M,K,device=dev); B=torch.randn(N,K,device=dev); out=torch.empty(M,N,device=dev)
A16,B16,out16=A.to(torch.bfloat16),B.to(torch.bfloat16),torch.empty(M,N,dtype=torch.bfloat16,device=dev)
for name, fn, fl in (('fp32 gemm_f32_mfma_kernel', lambda: ops.gemm(L.NT,A,B,out,M=M,N=N,K=K), 1), ('bf16 gemm_bf16_224p_kernel', lambda: ops.gemm(L.NT,A16,B16,out16,M=M,N=N,K=K), 1)):
    t0=time.time(); n=0
    torch.cuda.synchronize()
    while time.time()-t0 < 5:
        for _ in range(20): fn()
        torch.cuda.synchronize(); n+=20
    dt=time.time()-t0
    print(f'done {name}: {dt/n*1e6:.1f} us per launch = {2.0*M*N*K*n/dt/1e12:.1f} TFLOP/s', flush=True)
PY
PID=$!
sleep 2
for i in 1 2 3 4 5 6 7 8 9 10; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 1; done
wait $PID
