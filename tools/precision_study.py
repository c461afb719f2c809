#!/usr/bin/env python3
"""CPU study of the 16-bit error budget (dev tool; uses the oracle as the checker, nothing here ships).

Emulates the roundings of the HIP path one family at a time on the fp32 oracle arithmetic and reports, per weight
family and seed: logit std, mean / max |dlogit|, and the frame-label flip rate at threshold 0 -- raw and after the head
has been calibrated to logit std 2.

    python tools/precision_study.py --seconds 40 --chunk 10 --seeds 0 1 --family stress hubert
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import vtc_oracle as O  # noqa: E402
from vtc_amd import synth  # noqa: E402

ALL = ("conv", "proj", "qkv_a", "qkv_w", "qkv_s", "attn", "o_a", "o_w", "resid", "preln", "ff1_a", "ff1_w", "ff2_a", "ff2_w")


def h(x, on):
    return x.half().float() if on else x


@torch.no_grad()
def forward(w, wav, fam: set):
    """fam: the set of rounding families switched on (members of ALL)."""
    r = lambda name, t: h(t, name in fam)
    x = wav[:, None, :]
    x = F.conv1d(x, w["feature_extractor.conv_layers.0.conv.weight"], stride=5)
    x = F.group_norm(x, 512, w["feature_extractor.conv_layers.0.layer_norm.weight"],
                     w["feature_extractor.conv_layers.0.layer_norm.bias"], eps=1e-5)
    x = r("conv", F.gelu(x))
    for i in range(1, 7):
        x = F.conv1d(x, r("conv", w[f"feature_extractor.conv_layers.{i}.conv.weight"]), stride=2)
        x = r("conv", F.gelu(x))
    x = x.transpose(1, 2)
    x = F.layer_norm(x, (512,), w["feature_projection.layer_norm.weight"], w["feature_projection.layer_norm.bias"], eps=1e-5)
    x = F.linear(r("proj", x), r("proj", w["feature_projection.projection.weight"]), w["feature_projection.projection.bias"])
    pc = F.conv1d(r("proj", x.transpose(1, 2)), r("proj", O.pos_conv_weight(w)), w["encoder.pos_conv_embed.conv.bias"],
                  padding=64, groups=16)[:, :, :-1]
    x = x + F.gelu(pc).transpose(1, 2)
    x = F.layer_norm(x, (768,), w["encoder.layer_norm.weight"], w["encoder.layer_norm.bias"], eps=1e-5)
    b, t, _ = x.shape
    for l in range(12):
        p = f"encoder.layers.{l}."
        xa = r("qkv_a", x)
        xr = r("resid", x)
        qh = F.linear(xa, r("qkv_w", w[p + "attention.q_proj.weight"]), w[p + "attention.q_proj.bias"]) * 0.125
        kh = F.linear(xa, r("qkv_w", w[p + "attention.k_proj.weight"]), w[p + "attention.k_proj.bias"])
        vh = F.linear(xa, r("qkv_w", w[p + "attention.v_proj.weight"]), w[p + "attention.v_proj.bias"])
        qh, kh, vh = (r("qkv_s", z).view(b, t, 12, 64).transpose(1, 2) for z in (qh, kh, vh))
        att = torch.softmax(qh @ kh.transpose(2, 3), dim=-1)
        ctx = (r("attn", att) @ vh).transpose(1, 2).reshape(b, t, 768)
        u = xr + F.linear(r("o_a", ctx), r("o_w", w[p + "attention.out_proj.weight"]), w[p + "attention.out_proj.bias"])
        x = F.layer_norm(r("preln", u), (768,), w[p + "layer_norm.weight"], w[p + "layer_norm.bias"], eps=1e-5)
        xa = r("ff1_a", x)
        xr = r("resid", x)
        hh = F.gelu(F.linear(xa, r("ff1_w", w[p + "feed_forward.intermediate_dense.weight"]),
                             w[p + "feed_forward.intermediate_dense.bias"]))
        u = xr + F.linear(r("ff2_a", hh), r("ff2_w", w[p + "feed_forward.output_dense.weight"]),
                          w[p + "feed_forward.output_dense.bias"])
        x = F.layer_norm(r("preln", u), (768,), w[p + "final_layer_norm.weight"], w[p + "final_layer_norm.bias"], eps=1e-5)
    return x   # last hidden state; the head is applied by the caller (fp32)


def windows(audio, chunk):
    plan = O.plan_windows(len(audio), chunk)
    n = len(plan.starts) - 1          # full windows only
    return torch.from_numpy(np.stack([audio[s: s + plan.window] for s in plan.starts[:n]]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=40.0)
    ap.add_argument("--chunk", type=float, default=10.0)
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 1])
    ap.add_argument("--family", nargs="+", default=["stress", "hubert"])
    ap.add_argument("--sets", nargs="+", default=["all"], help="'all', 'each', or '+'-joined family lists, e.g. qkv_a+qkv_w; "
                    "a leading '-' means all except")
    args = ap.parse_args()
    torch.set_num_threads(8)
    for family in args.family:
        for seed in args.seeds:
            w = synth.make_weights(seed=seed, family=family)
            w = {k: torch.from_numpy(np.asarray(v)) for k, v in w.items()}
            audio = synth.synth_audio(args.seconds, seed=4000 + seed)
            x = windows(audio, args.chunk)
            ref_h = forward(w, x, set())
            hw, hb = w["head.weight"], w["head.bias"]
            ref = F.linear(ref_h, hw, hb).reshape(-1, 4)
            scale = 2.0 / float(ref.std())
            sets = []
            for s in args.sets:
                if s == "all":
                    sets.append(("all", set(ALL)))
                elif s == "each":
                    sets += [(a, {a}) for a in ALL]
                elif s.startswith("-"):
                    sets.append((s, set(ALL) - set(s[1:].split("+"))))
                else:
                    sets.append((s, set(s.split("+"))))
            for name, fam in sets:
                got_h = forward(w, x, fam)
                got = F.linear(got_h, hw, hb).reshape(-1, 4)
                d = (got - ref).abs()
                # calibrated head: weights scaled so that the oracle's logit std is 2 (the bias is kept)
                ref_c = F.linear(ref_h, hw * scale, hb).reshape(-1, 4)
                got_c = F.linear(got_h, hw * scale, hb).reshape(-1, 4)
                dc = (got_c - ref_c).abs()
                print(json.dumps({"family": family, "seed": seed, "set": name, "frames": int(ref.shape[0]),
                                  "logit_std": round(float(ref.std()), 3), "mean_err": round(float(d.mean()), 5),
                                  "max_err": round(float(d.max()), 5),
                                  "flip": round(float(((got >= 0) != (ref >= 0)).float().mean()), 6),
                                  "cal_std": round(float(ref_c.std()), 3), "cal_mean_err": round(float(dc.mean()), 5),
                                  "cal_max_err": round(float(dc.max()), 5),
                                  "cal_flip": round(float(((got_c >= 0) != (ref_c >= 0)).float().mean()), 6)}), flush=True)


if __name__ == "__main__":
    main()
