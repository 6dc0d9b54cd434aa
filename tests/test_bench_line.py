"""The bench line's contract with the driver's record (CPU only): the driver keeps the top-level scalars, `config`, and the FLAT scalars of `roofline` and
`cpu_baseline`; every nested object and every other key is dropped (BENCH_r05.json: extra_keys).  bench.flatten_for_driver repeats the figures the review
reads inside those three objects - checked here on the committed line of the final round-6 build (profiles/r06_bench.json), re-flattened from its nested
parts so that the function itself runs."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_flat_scalars_survive_the_drivers_parser():
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        import bench
    finally:
        sys.argv = argv
    line = json.load(open(os.path.join(ROOT, "profiles", "r06_bench.json")))
    raw = copy.deepcopy(line)
    for obj, keep in ((raw["roofline"], ("int_mad", "kernels")), (raw["cpu_baseline"], ())):       # strip what flatten added, keep the nested sources
        for k in [k for k in obj if k.startswith(("int_mad_", "ceiling_", "vgpr_spill", "ms_alone_", "blst_model", "value_over", "speedup_vs", "gpu_faster"))]:
            del obj[k]
    raw["config"] = {k: v for k, v in raw["config"].items() if k in ("workload", "global_batch", "blinding_chains", "parallelism", "batches_in_flight", "context_mode", "exchange")}
    bench.flatten_for_driver(raw)
    rf, cfg, cb = raw["roofline"], raw["config"], raw["cpu_baseline"]
    # what the driver's parser keeps: scalars only
    kept = {k: v for k, v in rf.items() if not isinstance(v, (dict, list))}
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "int_mad_frac", "int_mad_over_ceiling", "ceiling_tmads", "ceiling_clock_ghz",
              "ms_alone_k_hash_map", "ms_alone_k_hash_clear", "ms_alone_k_lines", "ms_alone_k_lineprod"):
        assert k in kept and kept[k] is not None, k
    assert 0.5 < kept["int_mad_frac"] < 0.75 and 0.7 < kept["int_mad_over_ceiling"] < 1.0
    assert abs(kept["int_mad_frac"] - rf["int_mad"]["frac"]) < 1e-12                      # the flat copy IS the nested figure
    for k in ("ms_one_caller", "msm_points_per_s", "msm_ms_per_call", "fav_32768_ms", "verify_one_signature_ms", "batch_64_ms", "batch_4096_ms",
              "latency_floor_ms", "build_stamp", "from_bytes_65536_ms"):
        assert cfg.get(k) is not None, k
    assert all(not isinstance(v, (dict, list)) or k == "exchange" for k, v in cfg.items())
    assert cfg["msm_points_per_s"] == raw["aux"]["g1_msm_2^20"]["points_per_s"] and cfg["fav_32768_ms"] == raw["aux"]["fastAggregateVerify_32768"]["ms_per_call"]
    assert cb["blst_model_vps"] == cb["cores"] / 400e-6 and cb["kind"] == "port" and cb["speedup_vs_cpu_port"] > 100
    # and the committed line already carries them (bench.py wrote it through the same function)
    for k in ("int_mad_frac", "ceiling_tmads"):
        assert line["roofline"][k] == rf[k]
    # round 6's latency marks on the box of the profile collection
    assert cfg["fav_32768_ms"] <= 3.2 and cfg["verify_one_signature_ms"] <= 3.2 and cfg["batch_64_ms"] <= 3.4 and cfg["batch_4096_ms"] <= 4.6 and cfg["latency_floor_ms"] <= 3.4
    assert cfg["ms_one_caller"] <= 12.6
