"""Host-side checks of bench.py's bookkeeping (no GPU): the contractual byte counts of SURVEY.md 8(d), the kernel byte
table, and the CPU baseline's shape."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_bytes_match_survey():
    b = _bench()
    assert b.algorithmic_bytes_per_cloud(8000, 64) == 10_013_952     # c2 / c5
    assert b.algorithmic_bytes_per_cloud(16000, 128) == 76_716_544   # c4
    # 60 % of 8 TB/s at c2 = the target quoted in BASELINE.json's north star
    assert abs(0.6 * 8.0e12 / b.algorithmic_bytes_per_cloud(8000, 64) - 479_332) < 2


def test_kernel_byte_table_is_below_the_contract():
    """This implementation moves fewer bytes than A(N,G) (two grids of the reference chain never reach HBM)."""
    b = _bench()
    kb = b.kernel_bytes_per_cloud(8000, 64)
    fused_step = kb["k_locate"] + kb["k_splat_hw"] + kb["k_zcol_fwdbwd"] + kb["k_gather_hw"]
    assert fused_step < b.algorithmic_bytes_per_cloud(8000, 64)
    assert set(b.CONFIGS) == {"c2", "c4", "c5"} and b.CONFIGS["c2"][:3] == (32, 8000, 64)
    # the point-record chunk is 256 x (16 + 16) bytes + bin offsets: the locate kernel writes more than it reads
    assert kb["k_locate"] > 12 * 8000 + 32 * 8000
    # the unfused column kernels do not write a `smoothed` grid on the hot path
    assert kb["k_zcol_fwd"] == 4 * 64 ** 3 + 4 * 64 ** 2 and kb["k_zcol_bwd"] == 8 * 64 ** 3 + 8 * 64 ** 2


def test_candidate_step_bytes_exclude_the_losers_backward():
    b = _bench()
    a = b.algorithmic_bytes_per_cloud(8000, 64)
    assert b.step_bytes_per_cloud(8000, 64, 1) == a
    fwd = 12 * 8000 + 16 * 64 ** 3 + 4 * 64 ** 2
    assert abs(b.step_bytes_per_cloud(8000, 64, 8) - (fwd + (a - fwd) / 8)) < 1e-6


def test_recorded_lines_stay_below_the_hbm_peak():
    """No per-kernel or whole-step rate in a bench line committed under profiles/ (this round's) may exceed the 8 TB/s
    peak: a figure above it means the byte model credits traffic the kernel does not have."""
    import glob
    import json

    b = _bench()
    seen = 0
    for path in glob.glob(os.path.join(ROOT, "profiles", "r0[234]_*.json*")):
        for line in open(path):
            line = line.strip()
            if not line.startswith("{"):
                continue
            try:
                rec = json.loads(line)
            except ValueError:
                continue
            rec = rec.get("line", rec) if isinstance(rec, dict) else rec
            if not isinstance(rec, dict) or "kernels_gbs" not in rec:
                continue
            seen += 1
            for k, v in rec["kernels_gbs"].items():
                assert v <= b.HBM_PEAK_GBS, (path, k, v)
            assert rec["roofline_step"]["frac"] <= 1.0, path
            if rec.get("roofline"):
                assert rec["roofline"]["frac"] <= 1.0, path
    assert seen >= 5


def test_gpus_flag_launches_ranks_before_touching_a_gpu(monkeypatch):
    """`python bench.py --gpus N` with no launcher: refuses loudly when the node has fewer devices, and otherwise builds a
    torch.distributed.run command for N ranks on 127.0.0.1."""
    import subprocess
    import types

    b = _bench()
    args = types.SimpleNamespace(gpus=4, rehearse_on_one_gpu=False)
    monkeypatch.setattr(b.torch.cuda, "device_count", lambda: 1)
    assert b.launch_ranks(args) == 2
    calls = []
    monkeypatch.setattr(b.torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(subprocess, "run", lambda cmd, env=None: calls.append((cmd, env)) or types.SimpleNamespace(returncode=0))
    monkeypatch.setattr(b.sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    assert b.launch_ranks(args) == 0
    cmd, env = calls[0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "7"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_measured_traffic_reads_the_committed_profile():
    b = _bench()
    t = b.measured_traffic("k_splat_xl") or b.measured_traffic("k_splat_hw")
    assert t is None or 1e6 < t < 1e9


def test_committed_headline_line_carries_the_contract_fields():
    """The first line of profiles/r04_bench_lines.jsonl is `python bench.py --steps 20 --warmup 5` as the driver runs it:
    every field of the bench contract, the roofline and CPU-baseline objects, and the launch settings that shape `value`."""
    import json

    with open(os.path.join(ROOT, "profiles", "r04_bench_lines.jsonl")) as fh:
        rec = json.loads(fh.readline())["line"]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in rec, key
    assert rec["steps"] == 20 and rec["warmup"] == 5 and rec["n_gpus"] == 1 and rec["vs_baseline"] is None
    assert abs(rec["value"] - 32 * 1e3 / rec["ms_per_step"]) < 1e-6 * rec["value"]
    roof = rec["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert roof["traffic"] and roof["traffic"] > 0.9 * roof["algorithmic_bytes_per_launch"]
    # the object names ONE kernel, chosen from a committed profile (not from this run's level event timings), and carries
    # every kernel of the step beside it
    assert roof["chosen_by"].startswith("profiles/") and set(roof["all_kernels"]) == set(rec["kernels_us"])
    assert all(0 < k["frac"] <= 1 for k in roof["all_kernels"].values())
    step = rec["roofline_step"]
    assert 0 < step["own_traffic_frac"] < step["frac"] <= 1.0    # what the launches really move vs the contractual A(N,G)
    cpu = rec["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and "32 clouds" in cpu["sample"]
    cfg = rec["config"]
    assert "model" not in cfg and cfg["workload"].startswith("BASELINE configs[1]")
    # the headline is one native call per step: no HIP graph, so no runtime graph setting shapes it
    assert "native call" in cfg["launch"] and cfg["hip_graph_packet_capture"] is None
    assert cfg["batches_in_flight"] == 1 and cfg["split"] == 1
    assert rec["two_batches_in_flight"]["point_clouds_per_sec"] > rec["value"]   # reported beside, never as, the value
    assert rec["hip_graph_replay"]["point_clouds_per_sec"] > 0
    # ... and so is the reference's own call sequence through the drop-in signatures, eager and replayed
    assert 0 < rec["plain_eager"]["point_clouds_per_sec"] < rec["plain_graph_replay"]["point_clouds_per_sec"] < rec["value"]


def test_dominant_kernel_is_chosen_from_the_committed_profile():
    """Two kernels of the c2 step are level (17.0 / 17.6 us) and used to swap places in the `roofline` object from run to run:
    the choice now comes from the committed rocprofv3 summary of the config."""
    b = _bench()
    a, src_a = b.dominant_kernel({"k_splat_xl": 0.0180, "k_gather_hw": 0.0170, "k_locate": 0.007, "k_zcol_fwdbwd": 0.015}, "c2")
    c, src_c = b.dominant_kernel({"k_splat_xl": 0.0170, "k_gather_hw": 0.0180, "k_locate": 0.007, "k_zcol_fwdbwd": 0.015}, "c2")
    assert a == c and src_a == src_c and src_a.startswith("profiles/")
    d, src_d = b.dominant_kernel({"k_unknown": 1.0, "k_other": 2.0}, "c2")
    assert d == "k_other" and "event" in src_d
