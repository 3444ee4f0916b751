import os, sys, time, numpy as np, torch
sys.path.insert(0, '.')
from ocrd_keraslm_amd.lib import hipabi
from ocrd_keraslm_amd.lib.engine import HipLM
L,W,V,B,T = int(os.environ.get("KL_PROBE_L", "2")),int(os.environ.get("KL_PROBE_W", "512")),256,int(sys.argv[1]) if len(sys.argv)>1 else 64,256
lm = HipLM(L,W,V,1)
lm.init_weights(seed=1)
lm.prepare(hipabi.KL_PREC_BF16)
rng=np.random.default_rng(0)
idx=torch.from_numpy(rng.integers(1,V,(B,T)).astype(np.int32)).cuda()
ctx=torch.from_numpy(rng.integers(0,200,(B,1,1)).repeat(T,axis=1).astype(np.int32)).cuda()
tgt=torch.from_numpy(rng.integers(1,V,(B,T)).astype(np.int32)).cuda()
masks=torch.from_numpy(lm.draw_dropout_masks(B)).cuda()
lm.reset_states(B)
def step():
    lm.train_window(idx,ctx,tgt,masks); lm.adam_step()
for _ in range(int(os.environ.get('KL_PROBE_WARM', '3'))): step()
torch.cuda.synchronize()
t=time.time(); n=int(os.environ.get('KL_PROBE_N', '10'))
for _ in range(n): step()
torch.cuda.synchronize()
dt=(time.time()-t)/n
print(f"train B={B} T={T}: {dt*1e3:.2f} ms/step, {B*T/dt/1e6:.2f} Mchars/s, loss", lm.read_loss())
if os.environ.get('KL_PROBE_TRACE'):
    import ctypes as C
    hipabi.check(lm.lib.kl_trace_enable(lm.handle, 1))
    for _ in range(3): step()
    torch.cuda.synchronize()
    for kind in (0, 1):
        nn, ms, pers, fl = C.c_int(), C.c_float(), C.c_int(), C.c_double()
        hipabi.check(lm.lib.kl_trace_read(lm.handle, kind, C.byref(nn), C.byref(ms), C.byref(pers), C.byref(fl)))
        print("  %s: %.3f ms per launch (%d launches), %.1f%% of the MFMA roof" % (
            lm.lib.kl_trace_kernel_name(lm.handle, kind).decode(), ms.value / max(nn.value, 1), nn.value,
            100 * fl.value / (ms.value / max(nn.value, 1) * 1e-3) / 2.5e15 if ms.value else 0))
    hipabi.check(lm.lib.kl_trace_enable(lm.handle, 0))
if os.environ.get('KL_PROBE_TRAIN_ONLY'): sys.exit(0)
# forward only (inference split)
lm.prepare(hipabi.KL_PREC_SPLIT)
for Bf in (1, B):
    lm.reset_states(Bf)
    i2=idx[:Bf].contiguous(); c2=ctx[:Bf].contiguous()
    for _ in range(2): lm.forward_window(i2,c2)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(5): lm.forward_window(i2,c2)
    torch.cuda.synchronize(); dt=(time.time()-t)/5
    print(f"forward split B={Bf}: {dt*1e3:.2f} ms/window, {Bf*T/dt/1e6:.3f} Mchars/s")
# incremental
for n_h in (128, 1024):
    lm.ensure_pool(2*n_h)
    a=torch.arange(n_h,dtype=torch.int32).cuda(); b=a+n_h
    ii=torch.from_numpy(rng.integers(1,V,n_h).astype(np.int32)).cuda(); cc=torch.zeros((n_h,1),dtype=torch.int32).cuda()
    for prec in (hipabi.KL_PREC_SPLIT, hipabi.KL_PREC_BF16):
        lm.prepare(prec)
        for _ in range(5): lm.step_slots(ii,cc,a,b); a,b=b,a
        torch.cuda.synchronize(); t=time.time(); k=100
        for _ in range(k): lm.step_slots(ii,cc,a,b); a,b=b,a
        torch.cuda.synchronize(); dt=(time.time()-t)/k
        print(f"step n={n_h} prec={prec}: {dt*1e6:.1f} us/step, {n_h/dt/1e6:.2f} M hyp-chars/s")
