"""Dev tool: per-stage error of the HIP path vs the CPU oracle (run on the GPU box)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from oracle import vtc_oracle as O
from vtc_amd.engine import VtcEngine

def main():
    chunk = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dt = sys.argv[3] if len(sys.argv) > 3 else "f16"
    w = O.make_weights(seed=0)
    W = int(round(chunk * 16000)); T = O.conv_out_len(W)
    rng = np.random.default_rng(1)
    wav = (rng.standard_normal((B, W), dtype=np.float32) * 0.1)
    wav[-1, W // 2:] = 0
    st = O.encoder_stages(w, torch.from_numpy(wav))
    eng = VtcEngine(w, 4, chunk, max_batch=B, mfma_dtype=dt)
    arena = torch.from_numpy(wav.reshape(-1)).cuda()
    start = np.arange(B) * W; valid = np.full(B, W); valid[-1] = W // 2
    dst = np.arange(B) * T; keep = np.full(B, T)
    logits = torch.zeros((B * T, 4), device="cuda")
    names = [f"conv{i}" for i in range(7)] + ["proj", "enc_in"] + [f"layer{l}" for l in range(12)]
    Ls = []; n = W
    for k, s in zip(O.CONV_KERNELS, O.CONV_STRIDES):
        n = (n - k) // s + 1; Ls.append(n)
    for stage, name in enumerate(names):
        eng.debug_stop_after(stage)
        eng.forward_windows(arena, start, valid, dst, keep, logits)
        ref = st[name]
        if stage <= 6:
            ref = ref.transpose(1, 2).contiguous()   # (B, L, 512) channel-last
        got = eng.debug_stage(stage, B, ref.numel())
        got = torch.from_numpy(got.reshape(ref.shape))
        d = (got - ref).abs()
        print(f"{name:8s} ref std {ref.std():.4f} maxabs {d.max():.5f} meanabs {d.mean():.6f}", flush=True)
    eng.debug_stop_after(-1)
    eng.forward_windows(arena, start, valid, dst, keep, logits)
    torch.cuda.synchronize()
    ref = st["logits"].reshape(B * T, 4)
    d = (logits.cpu() - ref).abs()
    print(f"logits   ref std {ref.std():.4f} maxabs {d.max():.5f} meanabs {d.mean():.6f} agree {((logits.cpu()>0)==(ref>0)).float().mean():.5f}")

main()
