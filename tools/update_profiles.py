"""Copy the outputs of tools/collect_profiles.sh (+ the MFMA PMC pass) from gpurun_out/r01 into profiles/ (the committed,
judged evidence) and derive the per-kernel traffic / MFMA-utilisation tables.   python tools/update_profiles.py [round]"""
import json, os, shutil, sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = os.path.join(R, "gpurun_out", "r01"), os.path.join(R, "profiles")


def last_line(p):
    return open(p).read().strip().splitlines()[-1]


for name in ("bench_default", "bench_under_rocprof"):
    with open(os.path.join(dst, f"{rnd}_{name}.json"), "w") as f:
        f.write(last_line(os.path.join(src, name + ".json")) + "\n")
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, f"{rnd}_bench_kernel_stats.csv"))

old = json.load(open(os.path.join(dst, f"{rnd}_pmc_traffic.json")))
notes = {k: v["note"] for k, v in old["kernels"].items() if "note" in v}
f, w = json.load(open(os.path.join(src, "pmc_FETCH_SIZE.json"))), json.load(open(os.path.join(src, "pmc_WRITE_SIZE.json")))
out = {"_how": old["_how"], "kernels": {}}
for k in sorted(set(f) | set(w)):
    a, b = f.get(k, {}).get("FETCH_SIZE", {}), w.get(k, {}).get("WRITE_SIZE", {})
    e = {"dispatches": a.get("dispatches") or b.get("dispatches"), "FETCH_SIZE_KB_avg": round(a.get("avg", 0), 1),
         "WRITE_SIZE_KB_avg": round(b.get("avg", 0), 1),
         "traffic_bytes_per_launch_raw": int((a.get("avg", 0) + b.get("avg", 0)) * 1024),
         "traffic_bytes_per_launch": int((2 * a.get("avg", 0) + b.get("avg", 0)) * 1024)}
    if k in notes:
        e["note"] = notes[k]
    out["kernels"][k] = e
json.dump(out, open(os.path.join(dst, f"{rnd}_pmc_traffic.json"), "w"), indent=1)

mp = os.path.join(src, "pmc_mfma.json")
if os.path.exists(mp):
    d = json.load(open(mp))
    oldm = json.load(open(os.path.join(dst, f"{rnd}_pmc_mfma.json")))
    om = {"_how": oldm["_how"], "kernels": {}}
    for k, v in sorted(d.items()):
        g = v.get("GRBM_GUI_ACTIVE", {}).get("avg", 0)
        if not g:
            continue
        mb, mops, n = v["SQ_VALU_MFMA_BUSY_CYCLES"]["avg"], v["SQ_INSTS_VALU_MFMA_MOPS_F32"]["avg"], v["SQ_INSTS_VALU_MFMA_F32"]["avg"]
        om["kernels"][k] = {"dispatches": v["GRBM_GUI_ACTIVE"]["dispatches"], "GRBM_GUI_ACTIVE": round(g),
                            "SQ_VALU_MFMA_BUSY_CYCLES": round(mb), "SQ_BUSY_CU_CYCLES": round(v["SQ_BUSY_CU_CYCLES"]["avg"]),
                            "SQ_INSTS_VALU_MFMA_MOPS_F32": round(mops), "SQ_INSTS_VALU_MFMA_F32": round(n),
                            "gflop_per_launch": round(mops * 512 / 1e9, 2), "busy_cycles_per_mfma": round(mb / n, 1),
                            "MfmaUtil": round(mb / (g / 8 * 1024), 4), "kernel_us_at_2.4GHz": round(g / 8 / 2400, 1)}
    json.dump(om, open(os.path.join(dst, f"{rnd}_pmc_mfma.json"), "w"), indent=1)

b = json.loads(last_line(os.path.join(src, "bench_default.json")))
r = b["roofline"]
print(f"value {b['value']} img/s, {b['ms_per_step']} ms/step, sample {b['sample_ddpm1000_images_per_sec']} img/s ({b['sample_seconds']} s), "
      f"train {b['train_tflops']} TF ({b['train_frac_of_f32_peak']}), sample frac {b['sample_frac_of_f32_peak']}")
print(f"roofline {r['kernel']}: {r['achieved']} TF frac {r['frac']} avg {r['avg_launch_us']} us traffic {r['traffic']}; mfma total {r['all_mfma_kernels_ms']} ms")
print("cpu", b["cpu_baseline"]["value"], b["cpu_baseline"]["sample_ddpm1000_images_per_sec"])
