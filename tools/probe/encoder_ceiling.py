"""What the 40 % MFMA target on whisper-base encoder fwd+bwd (north_star) would need, priced with the round-6 yardstick (profiles/r6_yardstick.json,
one box, one process, random data): every MFMA launch of the encoder's forward + backward at B = 64 timed at the FASTER of {vendor hipBLASLt /
SDPA, this library's kernel with a PLAIN epilogue}, and everything else -- LayerNorm, GELU, residual adds, bias, LoRA products and their
gradients, dropout masks, casts -- priced at ZERO.  That sum is a floor no assembly of today's best kernels for these shapes gets under.
Runs on the CPU (reads the committed JSON); writes profiles/r6_encoder_ceiling.json."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
y = json.load(open(os.path.join(ROOT, "profiles", "r6_yardstick.json")))
g = {(r["M"], r["N"], r["K"]): r for r in y["gemm"]}


def best(M, N, K):
    r = g[(M, N, K)]
    c = {"vendor": min(r[k]["min_ms"] for k in ("vendor_nt", "vendor_nn") if r[k]), "ns": r["ns_gemm"]["min_ms"]}
    w = min(c, key=c.get)
    return c[w] * 1e3, w, c


att = y["attention"][0]
fwd_ns, bwd_ns = att["fwd"]["ns_attn_fwd"]["min_ms"] * 1e3, att["bwd"]["ns_attn_bwd (one pass)"]["bwd_only_ms"] * 1e3
fwd_v = min(v["min_ms"] for k, v in att["fwd"].items() if k.startswith("sdpa") and v and "math" not in k) * 1e3
bwd_v = min(v["bwd_only_ms"] for k, v in att["bwd"].items() if k.startswith("sdpa") and v and "math" not in k) * 1e3
M = 96000
layer = [("q|k|v", (M, 1536, 512)), ("out_proj", (M, 512, 512)), ("fc1", (M, 2048, 512)), ("fc2", (M, 512, 2048)),
         ("fc2 dgrad", (M, 2048, 512)), ("fc1 dgrad", (M, 512, 2048)), ("out_proj dgrad", (M, 512, 512)), ("q|k|v dgrad", (M, 512, 1536))]
rows, tot = [], 0.0
for name, shp in layer:
    us, who, c = best(*shp)
    rows.append({"launch": name, "shape": shp, "us": round(us, 1), "kernel": who, "vendor_us": round(c["vendor"] * 1e3, 1), "ns_us": round(c["ns"] * 1e3, 1)})
    tot += us
rows.append({"launch": "attention forward", "us": round(min(fwd_ns, fwd_v), 1), "kernel": "ns" if fwd_ns <= fwd_v else "vendor", "vendor_us": round(fwd_v, 1), "ns_us": round(fwd_ns, 1)})
rows.append({"launch": "attention backward", "us": round(min(bwd_ns, bwd_v), 1), "kernel": "ns" if bwd_ns <= bwd_v else "vendor", "vendor_us": round(bwd_v, 1), "ns_us": round(bwd_ns, 1)})
tot += min(fwd_ns, fwd_v) + min(bwd_ns, bwd_v)
layers_ms = 6 * tot / 1e3
# conv stem: forward conv1.0 (M 384 000, K 624), conv1.2 (M 192 000, K 1536: half the yardstick's 384 000-row launch), conv2 (M 96 000, K 1536);
# backward: the two stride-2 input gradients (same FLOPs as their forwards) + three weight gradients (ns_gemm_tn256: 1.07 ms per step in
# profiles/r5_a_bench_kernel_stats.csv; no vendor figure taken -- generous: 302 + 302 + 151 GFLOP at the best forward rate seen, 1.2 PFLOP/s)
c10 = best(384000, 512, 624)[0]
c12 = best(384000, 512, 1536)[0] / 2
c2 = best(96000, 512, 1536)[0]
stem_ms = (c10 + c12 + c2 + c12 + c2 + (302e9 + 302e9 + 151e9) / 1.2e15 * 1e6) / 1e3
enc_gflop = 213.37 * 64
out = {"source": "profiles/r6_yardstick.json", "per_layer_us": rows, "six_layers_ms": round(layers_ms, 3), "conv_stem_ms": round(stem_ms, 3),
       "floor_ms": round(layers_ms + stem_ms, 3), "encoder_algorithmic_gflop_B64": round(enc_gflop, 1),
       "mfma_frac_at_floor": round(enc_gflop * 1e9 / ((layers_ms + stem_ms) * 1e-3) / 2.5e15, 4),
       "ms_allowed_by_40_percent": round(enc_gflop * 1e9 / (0.4 * 2.5e15) * 1e3, 3),
       "note": "floor = best known kernel per MFMA launch, every other operation of the step free; the 40 % target allows LESS time than this floor"}
json.dump(out, open(os.path.join(ROOT, "profiles", "r6_encoder_ceiling.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
