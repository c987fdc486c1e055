"""Host-side helpers of bench.py: plan label -> kernel symbol (the roofline block picks the dominant kernel by SYMBOL)
and the choice of the newest sha-matched counter record."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_plan_labels_map_to_the_symbols_rocprof_prints():
    stats = os.path.join(ROOT, "profiles", "round5_unet_b192_kernel_stats.csv")
    names = [line.split('","')[0].strip('"') for line in open(stats).read().splitlines()[1:]]
    cases = {"k_gemm<10>": "k_gemm_dma<SdnBF16, 10, 4, 2, 0>", "k_gemm<10>/rp": "k_gemm_dma<SdnBF16, 10, 4, 2, 0>",
             "k_gemm<10>/ln1": "k_gemm_dma<SdnBF16, 10, 4, 2, 1>", "k_gemm<10>/ln2": "k_gemm_dma<SdnBF16, 10, 4, 2, 2>",
             "k_gemm<8>": "k_gemm_dma<SdnBF16, 8, 4, 2, 0>", "k_conv_slab<64>": "k_conv_slab<SdnBF16, 64>",
             "k_conv_slab<16>": "k_conv_slab<SdnBF16, 16>", "k_ffn320": "k_ffn320<SdnBF16>"}
    for label, symbol in cases.items():
        name, pat = bench.label_symbol(label, "BF16")
        assert name == symbol, (label, name)
        hits = [n for n in names if re.search(pat, n)]
        assert len(hits) == 1 and symbol in hits[0], (label, hits)          # exactly one row of the rocprof summary
    # two labels, one symbol: what the dominant-kernel selection sums
    assert bench.label_symbol("k_gemm<10>", "BF16")[0] == bench.label_symbol("k_gemm<10>/rp", "BF16")[0]
    # one label, two instantiations (self- / cross-attention query sets): a family pattern
    name, pat = bench.label_symbol("k_attn<40>", "BF16")
    assert len([n for n in names if re.search(pat, n)]) == 2
    assert bench.label_symbol("k_gemm<5>", "F16")[0] == "k_gemm_dma<SdnF16, 5, 2, 2|4, 0>"
    assert bench.label_symbol("k_gn_apply", "BF16") == ("k_gn_apply", None) and bench.label_symbol("k_gemm<10>x3", "BF16")[1] is None


def test_counter_records_are_reported_only_for_the_running_library(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    for rnd, sha in ((4, "aaa"), (5, "bbb"), (10, "ccc")):
        (prof / f"round{rnd}_traffic.json").write_text(json.dumps({"__meta__": {"libsdn_sha256": sha}, "k": {"launches": 1}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    rec, note = bench._pmc_records("traffic", "ccc")                          # newest by ROUND NUMBER (10 > 5), not by name order
    assert rec is not None and "round10_traffic.json" in note
    rec, note = bench._pmc_records("traffic", "bbb")
    assert rec is None and "another build" in note and "round10" in note
    assert bench._pmc_records("mfma_util", "ccc")[0] is None
