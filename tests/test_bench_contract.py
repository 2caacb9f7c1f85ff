"""The bench line the driver parses (task contract 4): schema of the committed round-2 line and agreement between its HIP-event numbers and
the committed rocprofv3 summaries / PMC passes of the same commands.  CPU only: reads profiles/, runs nothing."""
import csv
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def _line(name):
    with open(os.path.join(P, name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def _norm(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return re.sub(r"\([^()]*\)$", "", name)


def _stats(name):
    with open(os.path.join(P, name)) as f:
        return {_norm(r["Name"]): (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(f)}


def test_bench_line_schema():
    d = _line("r02_bench_default.json")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] in ("bf16x3/f32", "f32")
    assert d["higher_is_better"] is True and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 128 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3           # whole-job throughput of K timed steps
    for key in ("roofline", "roofline_largest_flops"):
        r = d[key]
        assert r["bound"] in ("hbm", "mfma") and r["unit"] == ("GB/s" if r["bound"] == "hbm" else "TFLOP/s")
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
        assert r["peak"] == (8000.0 if r["bound"] == "hbm" else (2500.0 if ("bx3" in r["kernel"] or "attn_core" in r["kernel"]) else 157.3))
        assert "selection_rule" in r and r["launches_per_step"] >= 1 and r["avg_launch_us"] > 0
    assert "TOTAL TIME" in d["roofline"]["selection_rule"] and "FLOPs" in d["roofline_largest_flops"]["selection_rule"]
    # the dominant-by-time kernel really is the largest entry of the per-kernel table, and the table carries both rooflines per kernel
    mf = [k for k in d["train_step_kernels"] if k["mfma_peak"]]
    assert d["roofline"]["kernel"] == max(mf, key=lambda k: k["ms"])["kernel"]
    assert d["roofline_largest_flops"]["kernel"] == max(mf, key=lambda k: k["gflop"])["kernel"]
    for k in d["train_step_kernels"]:                    # binding resource: the matrix pipe by EXECUTED work (3 MFMAs per split-precision product) vs HBM
        assert k["bound"] == ("hbm" if k["frac_hbm"] >= k["frac_mfma_executed"] else "mfma")
        assert k["frac"] == (k["frac_hbm"] if k["bound"] == "hbm" else k["frac_mfma"])
    hbm = {k["kernel"].split(" ")[0]: k for k in d["train_step_kernels"] if k["mfma_peak"] is None}
    assert {"groupnorm_fwd", "groupnorm_bwd", "adam_step"} <= set(hbm)                   # HBM-bound families: bytes / us / 8 TB/s
    assert all(0 < hbm[n]["frac_hbm"] < 1 for n in ("groupnorm_fwd", "groupnorm_bwd", "adam_step"))
    if d["dtype"] != "f32":                      # the exact-f32 arithmetic is timed in the same run, beside the headline
        assert d["exact_f32_mode"]["train_images_per_sec"] > 0 and d["exact_f32_mode"]["train_images_per_sec"] < d["value"]
        assert not any(k.endswith("_frac_of_f32_peak") for k in d)                       # no fraction against the wrong peak
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "batch 128" in c["sample"] and c["unit"] == d["unit"]
    assert d["sample_ddpm1000_images_per_sec"] > 0 and d["sample_eager_launches"]["images_per_sec"] > 0 and d["sample_hip_graph"] is True


def test_training_kernels_agree_with_the_training_only_rocprof_summary():
    """HIP-event averages of the plain bench run vs rocprofv3 --kernel-trace --stats over `bench.py --mode train` (training dispatches only)."""
    d = _line("r02_bench_default.json")
    st = _stats("r02_train_kernel_stats.csv")
    checked = 0
    for k in d["train_step_kernels"]:
        sym = k["kernel"].split("(+")[0]
        if sym not in st or k["avg_us"] < 40 or "(+" in k["kernel"] or sym.startswith(("conv3_bx3_kernel<8,", "conv3_bx3_kernel<4,")):
            continue                              # short kernels: the event pair's own overhead dominates; '(+x)' and the 8x8 / 4x4 convolutions
                                                  # (split over channel chunks + splitk_epilogue_kernel): several symbols per bracketed call
        prof_us = st[sym][1]
        # the rocprof summary also covers the exact-f32 leg of the run for kernels both arithmetics use; the split-precision symbols are unique to it
        if "bx3" in sym or "attn_core" in sym:
            assert abs(prof_us - k["avg_us"]) / prof_us < 0.10, (sym, prof_us, k["avg_us"])
            checked += 1
    assert checked >= 6
    r = d["roofline"]
    assert r["kernel"].split("(+")[0] in st


def test_pmc_tables_cover_the_roofline_kernels():
    d = _line("r02_bench_default.json")
    with open(os.path.join(P, "r02_pmc_traffic.json")) as f:
        tr = json.load(f)["kernels"]
    with open(os.path.join(P, "r02_pmc_mfma.json")) as f:
        mf = json.load(f)["kernels"]
    for key in ("roofline", "roofline_largest_flops"):
        sym = d[key]["kernel"].split("(+")[0]
        assert sym in tr and tr[sym]["traffic_bytes_per_launch"] > 0 and "hbm_gbs" in tr[sym], sym
        # the bench line quotes the PMC table committed BEFORE it ran; the table is then re-collected with the line: same kernel, two passes
        assert abs(d[key]["traffic"] - tr[sym]["traffic_bytes_per_launch"]) / tr[sym]["traffic_bytes_per_launch"] < 0.05
    # MFMA-busy counter == executed flops / peak for an MFMA-bound kernel (a split-precision kernel executes 3 bf16 MFMAs per algorithmic
    # product term).  The counter ratio is per CYCLE, the bench's fraction per SECOND against the 2.4 GHz peak: convert with the kernel's
    # own cycles (GRBM) / measured duration.
    sym = "conv3_bx3_kernel<32, 1, 2, 512, 2>"            # stride-1 input gradient at 32x32, the 128 x 256 tile
    k = next(k for k in d["train_step_kernels"] if k["kernel"] == sym)
    m = mf[sym]
    util_time = m["MfmaUtil"] * m["kernel_us_at_2.4GHz"] / m["avg_us"]
    assert abs(util_time - k["frac_mfma_executed"]) < 0.04, (m["MfmaUtil"], util_time, k["frac_mfma_executed"])
    # HBM-bound kernels now have a reproducible GB/s: bytes of the PMC pass / duration of the SAME dispatch population
    for sym in ("gn_fwd_reg_kernel<4>", "gn_bwd_reg_kernel<4, 256>", "gn_bwd_reg_kernel<6, 512>", "adam_kernel"):
        assert sym in tr and 0 < tr[sym]["hbm_gbs"] < 8000, (sym, tr.get(sym))


def test_sampler_summary_is_sampler_only():
    st = _stats("r02_sample_kernel_stats.csv")
    assert not any("wgrad" in s or "gn_bwd" in s or "adam" in s for s in st)              # no training dispatches in the sampler's summary
    assert any(s.startswith("conv3_bx3_kernel<32, 3,") for s in st) and any(s.startswith("attn_core_kernel") for s in st)
    assert not any(s.startswith("softmax_col") for s in st)                              # the fused attention core replaced the column softmax



# ---- round 3: the stdout line is short (the driver keeps a ~9.5 KB stdout tail; round 2's 20.8 KB line arrived truncated), carries the parity
# gates, and the per-kernel tables live in the side file of the same run ----
def test_r03_bench_line_is_short_and_self_proving():
    raw = open(os.path.join(P, "r03_bench_default.json")).read().strip().splitlines()[-1]
    assert len(raw) < 4096, len(raw)
    d = json.loads(raw)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "parity"):
        assert k in d, k
    assert "train_step_kernels" not in d and "sampler_step_kernels" not in d            # those tables are what made the line too long
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "bf16x3/f32"
    assert "workload" in d["config"] and "model" not in d["config"] and "chunks of 128" in d["config"]["workload"]
    assert abs(d["value"] - 128 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    for k in ("kernel", "unit", "traffic", "launches_per_step", "avg_launch_us", "algorithmic_gflop_per_launch", "frac_executed"):
        assert k in r, k
    assert r["peak"] == 2500.0 and r["unit"] == "TFLOP/s" and r["kernel"].startswith("conv3_k32_kernel<32")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "batch 128" in c["sample"] and c["unit"] == d["unit"]
    p = d["parity"]                                    # SURVEY 8d gates, computed in the CPU leg on the same weights / batch
    assert p["pass"] is True and p["timestep_indices_bit_exact"] is True
    assert p["loss_step0_rel_err"] <= 1e-5 and p["grad_norm_rel_err"] <= 1e-4 and p["denoised_max_rel_err"] <= 1e-3
    assert d["exact_f32_mode"]["train_images_per_sec"] < d["value"] and d["sample_ddpm1000_images_per_sec"] > 0 and d["sample_hip_graph"] is True
    assert len(d["top_kernels"]) == 5 and d["top_kernels"][0]["kernel"] == r["kernel"]


def test_r03_detail_tables_agree_with_the_rocprof_summary_and_the_pmc_passes():
    d = _line("r03_bench_default.json")
    with open(os.path.join(P, "r03_bench_detail.json")) as f:
        det = json.load(f)
    assert det["value"] == d["value"] and det["roofline"]["kernel"] == d["roofline"]["kernel"] and "selection_rule" in det["roofline"]
    mf = [k for k in det["train_step_kernels"] if k["mfma_peak"]]
    assert d["roofline"]["kernel"] == max(mf, key=lambda k: k["ms"])["kernel"]
    for k in det["train_step_kernels"]:
        assert k["bound"] == ("hbm" if k["frac_hbm"] >= k["frac_mfma_executed"] else "mfma")
    st = _stats("r03_train_kernel_stats.csv")
    checked = 0
    for k in det["train_step_kernels"]:
        sym = k["kernel"].split("(+")[0]
        if sym not in st or k["avg_us"] < 40 or "(+" in k["kernel"] or sym.startswith(("conv3_bx3_kernel<8,", "conv3_bx3_kernel<4,")):
            continue
        if "bx3" in sym or "attn_core" in sym or "k32" in sym:
            # same code, two boxes of the pool (the bench line is re-taken after the PMC tables are committed): the boxes differ by a few per cent
            assert abs(st[sym][1] - k["avg_us"]) / st[sym][1] < 0.12, (sym, st[sym][1], k["avg_us"])
            checked += 1
    assert checked >= 6
    with open(os.path.join(P, "r03_pmc_traffic.json")) as f:
        tr = json.load(f)["kernels"]
    with open(os.path.join(P, "r03_pmc_mfma.json")) as f:
        mfm = json.load(f)["kernels"]
    sym = d["roofline"]["kernel"]
    assert sym in tr and tr[sym]["traffic_bytes_per_launch"] > 0 and tr[sym]["traffic_over_algorithmic"] < 1.5
    if d["roofline"]["traffic"] is not None:
        assert abs(d["roofline"]["traffic"] - tr[sym]["traffic_bytes_per_launch"]) / tr[sym]["traffic_bytes_per_launch"] < 0.05
    # MFMA-busy counter (per cycle) x the kernel's own clock == the bench's executed fraction (per second against the 2.4 GHz peak)
    k = next(k for k in det["train_step_kernels"] if k["kernel"] == sym)
    m = mfm[sym]
    util_time = m["MfmaUtil"] * m["kernel_us_at_2.4GHz"] / m["avg_us"]
    assert abs(util_time - k["frac_mfma_executed"]) < 0.05, (m["MfmaUtil"], util_time, k["frac_mfma_executed"])
    sst = _stats("r03_sample_kernel_stats.csv")
    assert not any("wgrad" in s or "gn_bwd" in s or "adam" in s for s in sst) and any(s.startswith("conv3_k32_kernel<32, 3>") for s in sst)


def _run_bench(args, env_extra, timeout=300):
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start 2 rank processes itself (before any GPU call) and rank 0 must
    print a line with n_gpus = 2 whose all-reduce counted 2 ranks.  Here on CPU: rendezvous + process-group proof only (gloo)."""
    r = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"VD_BENCH_BACKEND": "gloo", "VD_BENCH_RENDEZVOUS_ONLY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["process_group"]["ranks_counted_by_all_reduce"] == 2 and d["process_group"]["world_size"] == 2
    assert [x[0] for x in d["process_group"]["rank_device_pci"]] == [0, 1]


def test_gpus_flag_disagreeing_with_world_size_fails_loudly():
    r = _run_bench(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "VD_BENCH_RENDEZVOUS_ONLY": "1"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_no_split_precision_kernel_is_priced_against_the_f32_peak():
    """Round-4 review: `attn_flash_kernel` (bf16 x 3) was priced against the 157.3 TFLOP/s f32 peak (0.79 printed, 0.05 true).  The classification is
    now ONE table (villandiffusion_amd/flops.py) that raises for a matrix kernel nobody classified; every committed round-5 line must agree with it."""
    import glob
    from villandiffusion_amd.flops import is_split_precision
    for name in ("attn_flash_kernel<0>", "attn_flash_kernel<1>+<2>", "attn_core_kernel<8, true>", "conv3_k32p_kernel<32, 1, true, true, false>",
                 "gemm1x1_k32p_kernel<false>", "wgrad_k32_group_kernel<32, 0>(+group_reduce)", "wgrad1x1_wide_group_kernel(+group_reduce)",
                 "gemm_bx3_kernel<256>", "gemm_bx3_act_kernel<1, 0>", "wgrad9_group_kernel<32>(+group_reduce)", "wgrad_presplit_group_kernel<32, 0>"):
        assert is_split_precision(name), name
    for name in ("gemm_kernel<128x128,ROW,PLAIN>", "gemm_plain_kernel<1>", "conv3_patch_kernel<32, 0, 2>", "wgrad_kernel<128x128,CONV3_S2>(+slab_reduce)",
                 "wgrad_small_kernel<true>(+colsum)", "conv3_fewout_kernel<4>"):
        assert not is_split_precision(name), name
    with pytest.raises(KeyError):
        is_split_precision("brand_new_matrix_kernel<3>")
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    checked = 0
    for f in sorted(glob.glob(os.path.join(root, "r05_bench_*.json"))):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        rows = list(d.get("train_step_kernels") or []) + list(d.get("sampler_step_kernels") or [])
        for r in rows:
            if r.get("mfma_peak"):
                assert r["mfma_peak"] == (2500.0 if is_split_precision(r["kernel"]) else 157.3), (f, r["kernel"], r["mfma_peak"])
                checked += 1
        for key in ("roofline", "roofline_largest_flops"):
            r = d.get(key)
            if r and r.get("bound") == "mfma":
                assert r["peak"] == (2500.0 if is_split_precision(r["kernel"]) else 157.3), (f, key, r["kernel"], r["peak"])
                checked += 1
    print(f"[contract] {checked} priced kernel rows checked")


def test_round5_lines_carry_ddp_path_and_counted_traffic():
    """Round-4 review items 3 / 4: the default line reports the multi-rank schedule timed on one GPU (`ddp_path_ms_per_step`, within a few percent
    of the single-process step) and every committed round-5 line -- the default one and the secondary lines of BASELINE configs #4 / #5 -- carries
    the dominant kernel's HBM traffic from that configuration's own PMC passes, with a per-kernel traffic / algorithmic table beside it."""
    d = _line("r05_bench_default.json")
    assert d["metric"].startswith("train imgs/sec") and d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert abs(d["value"] - 128 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3
    assert d["ddp_path_ms_per_step"] and 0.95 <= d["ddp_path"]["vs_single_process"] <= 1.05
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] and 0.5 < r["traffic"] / (r["algorithmic_mbytes_per_launch"] * 1e6) < 3.0
    assert d["parity"]["pass"] is True and d["cpu_baseline"]["kind"] == "port"
    for tag in ("cfg4", "cfg5"):
        c = _line(f"r05_bench_{tag}.json")
        assert c["roofline"]["traffic"] and c["roofline"]["traffic_source"].startswith(f"profiles/r05_pmc_traffic_{tag}.json")
        with open(os.path.join(P, f"r05_pmc_traffic_{tag}.json")) as f:
            t = json.load(f)["kernels"]
        rows = [v for v in t.values() if "traffic_over_algorithmic" in v]
        assert len(rows) >= 10 and any("MfmaUtil" in v for v in t.values())
        k = c["roofline"]["kernel"].split("(+")[0].split("@")[0]
        assert t[k]["traffic_bytes_per_launch"] == c["roofline"]["traffic"]
    # the pre-split kernels are what the profiled step ran, and the counters see the matrix pipe busier than under the converting kernels
    with open(os.path.join(P, "r05_pmc_mfma.json")) as f:
        m = json.load(f)["kernels"]
    assert m["wgrad_ps_group_kernel<32, 0>"]["MfmaUtil"] > 0.6 and m["wgrad_ps_group_kernel<16, 0>"]["MfmaUtil"] > 0.6
