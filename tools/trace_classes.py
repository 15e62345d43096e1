#!/usr/bin/env python3
"""rocprofv3 per-class averages of the decode loop for bench.py's `roofline.classes`: parses a
`tools/trace_batch.sh` summary (gpurun_out/trace_<tag>_summary.md: kernel, calls, total ms, avg us) and writes
profiles/kernel_classes_b<B>.json = {class: {avg_us, calls, symbol}} + the hash of the kernel sources it was measured on.
bench.py attaches the numbers only when that hash equals the sources it runs (`csrc_sha()` there = the same function).
usage: trace_classes.py summary.md B [out.json]"""
import hashlib, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "plangen_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def classify(name):
    if "attn_decode_fused_kernel" in name: return "decode_attention"
    if "rmsnorm512_kernel" in name or "rmsnorm_kernel" in name or "rmsnorm_defer_kernel" in name: return "decode_rmsnorm"      # round 6: the deferred-1/rms form at 65..128 rows
    if "gemm_sk5_kernel" in name:
        m = re.search(r"gemm_sk5_kernel<(\d+), (\d+), (\d+), (\d+)", name)
        if m: return "decode_gemm_gate_up_swiglu" if int(m.group(4)) == 3 else {8: "decode_gemm_qkv", 16: "decode_gen_head_w2"}.get(int(m.group(1)))
    m = re.search(r"gemm_sk4_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)", name)
    if m:
        nck, epi = int(m.group(2)), int(m.group(5))
        if epi in (1, 3): return "decode_gemm_gate_up_swiglu"
        return {4: "decode_gemm_o", 11: "decode_gemm_down", 8: "decode_gemm_qkv", 16: "decode_gen_head_w2"}.get(nck)
    m = re.search(r"gemm_skinny3_kernel<(\d+), (\d+), (\d+), (\w+), (\d+)", name)
    if m:
        nck, epi = int(m.group(2)), int(m.group(5))
        if epi == 1: return "decode_gemm_gate_up_swiglu"
        return {4: "decode_gemm_o", 11: "decode_gemm_down", 8: "decode_gemm_qkv", 16: "decode_gen_head_w2"}.get(nck)
    if "cfg_scan_kernel" in name: return "decode_cfg_scan"
    if "cfg_pick_kernel" in name: return "decode_cfg_pick"
    return None


def main():
    md, B = sys.argv[1], int(sys.argv[2])
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", f"kernel_classes_b{B}.json")
    cls = {}
    for line in open(md):
        m = re.match(r"\| `(.*)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \|", line)
        if not m: continue
        name, calls, total_ms = m.group(1), int(m.group(2)), float(m.group(3))
        c = classify(name)
        if c is None or calls < 500: continue
        e = cls.setdefault(c, {"calls": 0, "total_ms": 0.0, "symbols": []})
        e["calls"] += calls; e["total_ms"] += total_ms; e["symbols"].append(name[:70])
    for e in cls.values():
        e["avg_us"] = e["total_ms"] * 1e3 / e["calls"]
    # one kernel INSTANTIATION can serve several GEMMs of a layer (64 rows: qkv, o and down all run gemm_sk4_kernel<4, 4, ..>): rocprofv3 cannot tell
    # them apart, so such a symbol's average is kept under a merged name that bench.py does not attach to any single class
    a_calls = cls.get("decode_attention", {}).get("calls", 0)
    for c in [c for c in cls if c.startswith("decode_gemm_")]:
        if a_calls and cls[c]["calls"] > 1.5 * a_calls:
            e = cls.pop(c)
            cls[c + "+merged_%dx" % round(e["calls"] / a_calls)] = e
    json.dump({"batch": B, "csrc_sha": csrc_sha(), "source": os.path.basename(md), "what": "rocprofv3 --kernel-trace --stats kernel durations of one bench step (tools/trace_batch.sh)",
               "classes": cls}, open(out, "w"), indent=1)
    print(json.dumps({k: round(v["avg_us"], 2) for k, v in cls.items()}))


if __name__ == "__main__":
    main()
