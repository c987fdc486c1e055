"""Driver-side formats: the JSON -> CLI -> YAML configuration order with the reference's quirks
(run_nudity.py:294-325,534-625) and the artefact names / keys it writes (run_nudity.py:249-262,466-529; main_utils.py)."""
import json
import os

import pytest
import yaml

from safe_denoiser_amd import driver


class Img:
    def __init__(self):
        self.saved = []

    def save(self, path):
        self.saved.append(path)
        open(path, "wb").write(b"png")


@pytest.fixture
def cfg_files(tmp_path):
    cfg = {"erase_id": "safree_neg_prompt", "safree": True, "svf": True, "lra": True, "guidance_scale": 7.5,
           "num_inference_steps": 50, "image_length": 512, "nudity": "nudity", "category": "IGNORED", "save_dir": str(tmp_path / "out")}
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    task = {"repellency": {"method": "kernel_fast", "n_embed": 16, "guidance_scale": 0.0,
                           "params": {"scale": 0.33, "sigma": 3.15, "proj_ref_path": "caches/sd/i2p_sexual/repellency_proj_ref.pt",
                                      "cache_proj_ref": True, "beta_threshold_margin": 1.6}},
            "data": {"name": "nudity", "root": "datasets/nudity", "class_info": "i2p_sexual"},
            "mean_processor": {"dynamic_threshold": False, "clip_denoised": True}}
    (tmp_path / "task.yaml").write_text(yaml.dump(task))
    return tmp_path


def test_three_layer_configuration_order_and_quirks(cfg_files):
    cj, ty = str(cfg_files / "cfg.json"), str(cfg_files / "task.yaml")
    a = driver.parse_args(["--config", cj])
    # layer 1: JSON supplies the defaults, including the svf / lra -> flag-name mapping and the "nudity" key for --category
    assert a.erase_id == "safree_neg_prompt" and a.safree and a.self_validation_filter and a.latent_re_attention
    assert a.category == "nudity" and a.save_dir == str(cfg_files / "out") and a.re_attn_t == "-1,1001" and a.up_t == 10
    # layer 2: the command line overrides it
    b = driver.parse_args(["--config", cj, "--erase_id=safree_neg_prompt_rep_threshold_time", "--guidance_scale", "5",
                           "--task_config", ty, "--save-dir", "/tmp/x", "--valid_case_numbers", "65,130"])
    assert b.erase_id == "safree_neg_prompt_rep_threshold_time" and b.guidance_scale == 5.0 and b.save_dir == "/tmp/x"
    with pytest.raises(SystemExit):
        driver.parse_args(["--config", cj, "--category", "artists-VanGogh"])      # choices are 'nudity' | 'all' (:581)
    # layer 3: YAML -- consumed keys vs parsed-and-ignored ones
    tc = driver.load_task_config(b.task_config)

    class Sch:
        betas, beta_start, beta_end = [0.0] * 1000, 0.00085, 0.012
    kw = driver.repellency_kwargs(tc, b.num_inference_steps, Sch())
    assert kw["name"] == "kernel_fast" and kw["n_embed"] == 16 and kw["max_idx"] == 1000 and kw["num_timesteps"] == 50
    assert kw["sigma"] == 3.15 and kw["scale"] == 0.33 and kw["beta_threshold_margin"] == 1.6 and kw["cache_proj_ref"] is True
    assert "guidance_scale" not in kw and "mean_processor" not in kw                 # read, never forwarded
    bad = dict(tc); bad.pop("mean_processor")
    (cfg_files / "bad.yaml").write_text(yaml.dump(bad))
    with pytest.raises(KeyError):
        driver.load_task_config(str(cfg_files / "bad.yaml"))
    assert driver.load_task_config(None) is None
    # erase_id registry and the derived call arguments
    assert driver.ERASE_IDS[b.erase_id] == ("safree", "threshold_time", True) and driver.ERASE_IDS["std"][1] is None
    space, neg = driver.negative_prompts(b)
    assert len(space) == 17 and neg == ", ".join(space)
    a.erase_id = "std"
    assert driver.negative_prompts(a) == ([" "], None)
    sf = driver.safree_dict(b)
    assert sf["re_attn_t"] == [-1, 1001] and sf["svf"] and sf["lra"] and sf["safree"] and sf["alpha"] == 0.01
    assert driver.SLD_CONFIGS["MEDIUM"]["sld_guidance_scale"] == 1000 and driver.SLD_CONFIGS["MAX"]["sld_warmup_steps"] == 0


def test_artifacts_tree_names_and_keys(cfg_files):
    cj, ty = str(cfg_files / "cfg.json"), str(cfg_files / "task.yaml")
    args = driver.parse_args(["--config", cj, "--task_config", ty])
    tc = driver.load_task_config(args.task_config)
    art = driver.RunArtifacts(args, tc)
    root = args.save_dir
    assert sorted(os.listdir(root)) == ["all", "logs.txt", "safe", "unsafe"]
    cases = [dict(prompt="p0", case_number=7, seed=1, guidance=7.5, categories=["sexual", "violence"], row=0),
             dict(prompt="p1", case_number=9, seed=2, guidance=7.5, categories="nudity", row=1)]
    verdicts = iter([(True, 0.91), (False, 0.12)])
    for c in cases:
        art.log_case(c)
        art.record(c, Img(), eval_func=lambda imgs, threshold: next(verdicts))
        art.log_time(c, 1.234)
    art.finish(dataset_size=2)
    assert os.path.exists(os.path.join(root, "unsafe", "7_sexual-violence.png"))
    assert os.path.exists(os.path.join(root, "all", "7_sexual-violence.png"))
    assert os.path.exists(os.path.join(root, "safe", "9_n-u-d-i-t-y.png"))          # a plain string joins its characters
    d = json.load(open(os.path.join(root, "detect_dict.json")))
    assert d["unsafe"] == [True, False]
    assert d["toxic_ratio"] == {"sexual": 1.0, "violence": 1.0, "nudity": 0.0, "average": 0.5}
    assert d["toxic_size"]["average"] == 2 and abs(d["toxic_pred_ratio"]["nudity"] - 0.12) < 1e-12
    merged = yaml.safe_load(open(os.path.join(root, "config.yaml")))
    assert merged["erase_id"] == "safree_neg_prompt" and merged["repellency"]["params"]["sigma"] == 3.15
    assert merged["mean_processor"]["clip_denoised"] is True and merged["self_validation_filter"] is True
    log = open(os.path.join(root, "logs.txt")).read()
    for needle in ("All configurations provided:", "erase_id: safree_neg_prompt", "Seed: 1, Iter: 0, Case#: 7: target prompt: p0",
                   "Optimized image is unsafe: True, toxicity pred: 0.910", "Wall-Clock Time for image generation (Case#: 9): 1.23 seconds",
                   "safe: 1, unsafe: 1", " - INFO - "):
        assert needle in log, needle


def test_rank_directories_do_not_collide(cfg_files):
    args = driver.parse_args(["--config", str(cfg_files / "cfg.json")])
    a0, a1 = driver.RunArtifacts(args, None, rank=0, world=2), driver.RunArtifacts(args, None, rank=1, world=2)
    assert a0.save_dir.endswith("rank00") and a1.save_dir.endswith("rank01") and a0.save_dir != a1.save_dir
    c = dict(prompt="p", case_number=3, seed=1, guidance=7.5, categories="nudity", row=0)
    a0.record(c, Img()); a1.record(dict(c, case_number=4), Img())
    a0.finish(); a1.finish()
    assert os.path.exists(os.path.join(a0.save_dir, "all", "3_n-u-d-i-t-y.png"))
    assert json.load(open(os.path.join(a1.save_dir, "detect_dict.json"))) == {}


class RecordingPipe:
    """Stands where SafeDenoiserPipeline sits in run_job: records every call's keyword arguments, returns one image per prompt."""

    def __init__(self, variant):
        self.variant, self.calls = variant, []

    def __call__(self, prompts, **kw):
        self.calls.append(dict(kw, prompts=list(prompts)))
        return [Img() for _ in prompts]


def _job(tmp_path, erase_id, extra=(), rows=5, guidance_col=False):
    head = "case_number,prompt,categories,evaluation_seed" + (",guidance" if guidance_col else "")
    lines = [head] + [f'{10 + i},"prompt {i}",sexual,{100 + i}' + (f",{7.5 if i % 2 else 9.0}" if guidance_col else "") for i in range(rows)]
    (tmp_path / "p.csv").write_text("\n".join(lines) + "\n")
    cfg = {"erase_id": erase_id, "nudity": "nudity", "data": str(tmp_path / "p.csv"), "save_dir": str(tmp_path / f"out_{erase_id}"),
           "safree": True, "svf": True, "lra": True}
    (tmp_path / f"{erase_id}.json").write_text(json.dumps(cfg))
    return driver.parse_args(["--config", str(tmp_path / f"{erase_id}.json"), *extra])


def test_only_the_rep_classes_receive_the_repellency_processor(cfg_files):
    """with --task_config the reference builds the processor for EVERY erase_id, but only the *_Rep classes read it
    (run_nudity.py:56-73; models/textuals/*.py never touch `repellency_processor`).  'rece' is ModifiedSLDPipeline with NO
    SafetyConfig splat ("sld" is not in its id, :329-334) -> the call's own defaults (models/textuals/modified_sld_pipeline.py:304-308)."""
    ty = str(cfg_files / "task.yaml")
    proc = object()
    seen = {}
    for eid in ("rece", "sld", "safree", "safree_neg_prompt", "std", "std_rep", "safree_neg_prompt_rep_threshold_time", "sld_rep_time"):
        args = _job(cfg_files, eid, ["--task_config", ty, "--safe_level", "STRONG"])
        pipe = RecordingPipe(driver.ERASE_IDS[eid][1])
        driver.run_job(args, pipe, proc, driver.load_task_config(ty), prompts_per_batch=4, device="cpu")
        seen[eid] = pipe.calls
    for eid in ("rece", "sld", "safree", "safree_neg_prompt", "std"):
        assert all(c["repellency_processor"] is None for c in seen[eid]), eid
    for eid in ("std_rep", "safree_neg_prompt_rep_threshold_time", "sld_rep_time"):
        assert all(c["repellency_processor"] is proc for c in seen[eid]), eid
    rece = seen["rece"][0]
    assert {k: rece[k] for k in driver.SLD_CALL_DEFAULTS} == dict(sld_guidance_scale=1000, sld_warmup_steps=10, sld_threshold=0.01,
                                                                  sld_momentum_scale=0.3, sld_mom_beta=0.4)
    assert "negation_warmup_steps" not in rece                                   # that key only comes with a SafetyConfig
    assert seen["sld"][0]["sld_guidance_scale"] == 2000 and seen["sld"][0]["negation_warmup_steps"] == 20      # STRONG, splatted
    assert seen["sld_rep_time"][0]["sld_warmup_steps"] == 7
    assert "sld_guidance_scale" not in seen["safree"][0] and "sld_guidance_scale" not in seen["std"][0]
    # negative prompt space: the 17 phrases only for ids containing "safree" (:345-371)
    assert len(seen["safree"][0]["negative_prompt_space"]) == 17 and seen["rece"][0]["negative_prompt_space"] == [" "]
    assert seen["safree"][0]["negative_prompt"] is None and seen["safree_neg_prompt"][0]["negative_prompt"].startswith("Sexual Acts, ")


def test_run_job_batches_keep_mixed_guidance_together_and_fold_the_tail(cfg_files):
    args = _job(cfg_files, "safree_neg_prompt_rep_threshold_time", rows=9, guidance_col=True)
    pipe = RecordingPipe("threshold_time")
    driver.run_job(args, pipe, None, None, prompts_per_batch=4, device="cpu")
    assert [len(c["prompts"]) for c in pipe.calls] == [4, 5]                      # 9 = 4 + (4 + 1): no one-prompt batch
    assert pipe.calls[0]["guidance_scale"] == [9.0, 7.5, 9.0, 7.5]                # per prompt, table order, one batch
    assert [g.initial_seed() for g in pipe.calls[1]["generator"]] == [104, 105, 106, 107, 108]


def test_merge_rank_outputs_rebuilds_the_single_tree_verdict(cfg_files):
    args = driver.parse_args(["--config", str(cfg_files / "cfg.json")])
    verdicts = {0: [(True, 0.9), (False, 0.2), (True, 0.8)], 1: [(False, 0.1), (True, 0.7)]}
    for r in (0, 1):
        art = driver.RunArtifacts(args, None, rank=r, world=2)
        for i, v in enumerate(verdicts[r]):
            c = dict(prompt="p", case_number=10 * r + i, seed=1, guidance=7.5, categories=["sexual"] if i % 2 == 0 else ["hate"], row=i)
            art.record(c, Img(), eval_func=lambda imgs, threshold, v=v: v)
        art.finish()
    m = driver.merge_rank_outputs(args.save_dir, 2)
    # TABLE order: cases are sharded r::2, so rank 0 holds table positions 0, 2, 4 and rank 1 positions 1, 3
    assert m["unsafe"] == [True, False, False, True, True]
    assert m["toxic_size"] == {"sexual": 3, "hate": 2, "average": 5}
    assert abs(m["toxic_ratio"]["sexual"] - 2 / 3) < 1e-12 and m["toxic_ratio"]["hate"] == 0.5 and abs(m["toxic_ratio"]["average"] - 0.6) < 1e-12
    assert abs(m["toxic_pred_ratio"]["sexual"] - (0.9 + 0.8 + 0.1) / 3) < 1e-12
    assert json.load(open(os.path.join(args.save_dir, "detect_dict.json"))) == m


def test_merge_rank_outputs_names_a_missing_rank_and_skips_an_empty_one(cfg_files):
    args = driver.parse_args(["--config", str(cfg_files / "cfg.json")])
    art = driver.RunArtifacts(args, None, rank=0, world=3)
    art.record(dict(prompt="p", case_number=1, seed=1, guidance=7.5, categories=["sexual"], row=0), Img(), eval_func=lambda imgs, threshold: (True, 0.9))
    art.finish()
    driver.RunArtifacts(args, None, rank=1, world=3).finish()                     # a rank whose shard was empty: {} -- skipped
    with pytest.raises(FileNotFoundError, match="rank 2 of 3"):
        driver.merge_rank_outputs(args.save_dir, 3)
    driver.RunArtifacts(args, None, rank=2, world=3).finish()
    m = driver.merge_rank_outputs(args.save_dir, 3)
    assert m["unsafe"] == [True] and m["toxic_size"] == {"sexual": 1, "average": 1}


class _U8Pipe:
    """A pipeline that can hand back uint8 image tensors (what SafeDenoiserPipeline does with a VAE attached): run_job then writes
    batch k on its worker thread while batch k + 1 is generated.  Deterministic images; logs a SAFREE line through the logger it is given."""
    variant = "threshold_time"

    class vae:
        @staticmethod
        def decode_latents_uint8(x):
            raise AssertionError("not called: the fake returns images itself")

    def __init__(self):
        self.calls = 0

    def __call__(self, prompts, output_type="pil", safree_dict=None, **kw):
        import numpy as np
        import torch
        from PIL import Image
        self.calls += 1
        for p in prompts:
            safree_dict["logger"].log(f"Among {len(p.split())} tokens, we remove 1.")
        arr = np.stack([np.full((8, 8, 3), (37 * self.calls + 11 * i) % 256, dtype=np.uint8) for i in range(len(prompts))])
        return torch.from_numpy(arr) if output_type == "uint8" else [Image.fromarray(a) for a in arr]


def _tree(root):
    out = {}
    for d, _, files in os.walk(root):
        for f in files:
            data = open(os.path.join(d, f), "rb").read()
            if f == "logs.txt":                                                   # drop the timestamps and the wall-clock figures
                import re
                data = re.sub(rb"^\S+ \S+ - ", b"", data, flags=re.M)
                data = re.sub(rb"\): [0-9.]+ seconds", b"): T seconds", data)
                data = re.sub(rb"out_overlap_(True|False)", b"out_overlap_X", data)           # (the configuration dump names save_dir)
            if f == "config.yaml":
                continue                                                          # (holds save_dir)
            out[os.path.relpath(os.path.join(d, f), root)] = data
    return out


def test_overlapped_host_io_writes_exactly_the_serial_tree(cfg_files):
    """PNG encodes + classifier + log lines of batch k run on a worker thread while batch k + 1 is generated;
    logs.txt (line order included), detect_dict.json and every image file must equal the serial loop's."""
    verdict = lambda imgs, threshold: (imgs[0].getpixel((0, 0))[0] % 2 == 0, imgs[0].getpixel((0, 0))[0] / 255.0)
    trees, timings = [], []
    for overlap in (False, True):
        args = _job(cfg_files, "safree_neg_prompt_rep_threshold_time", rows=11)
        args.save_dir = str(cfg_files / f"out_overlap_{overlap}")
        t = {}
        driver.run_job(args, _U8Pipe(), None, None, eval_func=verdict, prompts_per_batch=4, device="cpu", overlap_io=overlap, max_overfill=0.0,
                       timings=t)
        trees.append(_tree(args.save_dir)); timings.append(t)
    assert [b["prompts"] for b in timings[1]["batches"]] == [4, 4, 3]             # max_overfill = 0: the cap is strict
    assert trees[0].keys() == trees[1].keys() and len(trees[0]) >= 2 + 2 * 11     # logs, detect_dict, all/ + safe|unsafe/ per case
    for k in trees[0]:
        assert trees[0][k] == trees[1][k], k
    assert b"Among 2 tokens, we remove 1." in trees[1]["logs.txt"] and timings[1]["host_io_s"] > 0.0


def test_a_failing_writer_fails_the_job(cfg_files):
    def bad(imgs, threshold):
        raise RuntimeError("classifier down")
    # more batches than the writer keeps in flight: the producer must not wait forever for slots a dead worker never returns
    args = _job(cfg_files, "safree_neg_prompt_rep_threshold_time", rows=24)
    args.save_dir = str(cfg_files / "out_failing")
    with pytest.raises(RuntimeError, match="classifier down"):
        driver.run_job(args, _U8Pipe(), None, None, eval_func=bad, prompts_per_batch=4, device="cpu", overlap_io=True)
