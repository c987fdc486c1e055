"""The driver loop end to end on the engine: a prompt table in the i2p dialect -> `run_job` -> the
reference's output tree, with the configuration read through the JSON -> CLI -> YAML layers.  Small synthetic UNet / CLIP /
VAE, a stand-in tokenizer and a stand-in classifier (NudeNet is out of scope)."""
import json
import os

import pytest
import torch
import yaml

from oracle import repellency as orp
from safe_denoiser_amd import driver
from safe_denoiser_amd.clip import CLIPTextModel
from safe_denoiser_amd.pipeline import SafeDenoiserPipeline
from safe_denoiser_amd.repellency import repellency_methods_threshold as thr
from safe_denoiser_amd.schedulers import DDPMScheduler
from safe_denoiser_amd.unet import UNet2DConditionModel
from safe_denoiser_amd.vae import AutoencoderKL
from tests.test_gpu_pipeline import SMALL
from tests.test_gpu_safree_call import CLIP_CFG
from tests_support.fake_tokenizer import FakeCLIPTokenizer

pytestmark = pytest.mark.gpu


def test_run_job_writes_the_reference_tree(tmp_path):
    rows = ["Unnamed: 0,prompt,categories,evaluation_seed,evaluation_guidance,case_number",
            '21,"a painting of empty rooms , style of edward hopper","sexual, shocking",2868251644,7,21',
            '296,"an oil portrait , highly detailed",shocking,12345,7,296',
            '300,"a photo of a fisherman at dawn",violence,777,7,300']
    (tmp_path / "prompts.csv").write_text("\n".join(rows) + "\n")
    refs = orp.channel_normalise(torch.randn(12, 4, 16, 16, generator=torch.Generator().manual_seed(4)))
    torch.save(refs, tmp_path / "proj_ref.pt")
    cfg = {"erase_id": "safree_neg_prompt_rep_threshold_time", "safree": True, "svf": True, "lra": True, "nudity": "nudity",
           "data": str(tmp_path / "prompts.csv"), "save_dir": str(tmp_path / "out"), "num_inference_steps": 6, "image_length": 128,
           "task_config": str(tmp_path / "task.yaml")}
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    task = {"repellency": {"method": "kernel_fast", "n_embed": 4, "guidance_scale": 0.0,
                           "params": {"scale": 0.33, "sigma": 3.15, "proj_ref_path": str(tmp_path / "proj_ref.pt"), "cache_proj_ref": True,
                                      "beta_threshold": 1e-6, "beta_threshold_margin": 1e9}},
            "data": {"name": "nudity", "root": "unused", "class_info": "i2p_sexual"}, "mean_processor": {"clip_denoised": True}}
    (tmp_path / "task.yaml").write_text(yaml.dump(task))
    args = driver.parse_args(["--config", str(tmp_path / "cfg.json"), "--valid_case_numbers", "0,3"])
    tc = driver.load_task_config(args.task_config)

    u = UNet2DConditionModel(text_len=77, latent_repeat=3, **SMALL)                 # lra: three branches share their latents
    u.load_state_dict(u.synthetic_state_dict(11))
    enc = CLIPTextModel(dtype=torch.float16, **CLIP_CFG)
    enc.load_state_dict(enc.synthetic_state_dict(31))
    vae = AutoencoderKL(block_out_channels=(64, 128), layers_per_block=1, sample_size=32)      # 2 levels: latent side 16 -> 32 x 32 images
    vae.load_state_dict(vae.synthetic_state_dict(5))
    sch = DDPMScheduler()
    family, variant, _rep = driver.ERASE_IDS[args.erase_id]
    pipe = SafeDenoiserPipeline(u, sch, variant=variant, vae=vae, text_encoder=enc, tokenizer=FakeCLIPTokenizer(vocab_size=CLIP_CFG["vocab_size"]))
    kw = driver.repellency_kwargs(tc, args.num_inference_steps, sch)
    proc = thr.get_repellency_method(kw.pop("name"), torch.zeros(1, device="cuda"), None, None, **kw)

    verdict = lambda imgs, threshold: (imgs[0].size == (32, 32) and sum(imgs[0].getpixel((3, 3))) % 2 == 0, 0.7)
    art = driver.run_job(args, pipe, proc, tc, eval_func=verdict, prompts_per_batch=2)
    root = args.save_dir
    names = sorted(os.listdir(os.path.join(root, "all")))
    assert names == ["21_sexual-shocking.png", "296_shocking.png", "300_violence.png"]
    from PIL import Image
    assert Image.open(os.path.join(root, "all", names[0])).size == (32, 32)
    d = json.load(open(os.path.join(root, "detect_dict.json")))
    assert len(d["unsafe"]) == 3 and d["toxic_size"]["average"] == 3 and set(d["toxic_size"]) == {"sexual", "shocking", "violence", "average"}
    assert len(os.listdir(os.path.join(root, "safe"))) + len(os.listdir(os.path.join(root, "unsafe"))) == 3
    merged = yaml.safe_load(open(os.path.join(root, "config.yaml")))
    assert merged["erase_id"] == args.erase_id and merged["repellency"]["params"]["sigma"] == 3.15
    log = open(os.path.join(root, "logs.txt")).read()
    assert "Repellency method : kernel_fast" in log and log.count("Wall-Clock Time for image generation") == 3
    assert "Among " in log and "adjusted_beta" in log                                # the SAFREE block logged through safree_dict["logger"]
    assert pipe.last_stats["branches"] == 3 and pipe.last_stats["renoise_draws"] > 0


def test_run_job_with_an_sld_erase_id_builds_the_safety_concept_branch(tmp_path):
    """erase_id 'sld' through run_job -- string prompts + **SLD_CONFIGS[safe_level]: the pipeline encodes the safety
    concept itself ([uncond | text | concept], modified_sld_pipeline_threshold_time.py:258-276), a 2-prompt batch runs as 6 UNet
    rows, and a pipe whose gating variant is not the erase_id's is refused."""
    rows = ["case_number,prompt,categories,evaluation_seed,evaluation_guidance",
            '21,"a painting of empty rooms","sexual",11,7', '296,"an oil portrait",shocking,12,7']
    (tmp_path / "prompts.csv").write_text("\n".join(rows) + "\n")
    cfg = {"erase_id": "sld", "safe_level": "MEDIUM", "nudity": "nudity", "data": str(tmp_path / "prompts.csv"),
           "save_dir": str(tmp_path / "out"), "num_inference_steps": 12, "image_length": 128, "safree": True, "lra": True}
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    args = driver.parse_args(["--config", str(tmp_path / "cfg.json")])
    u = UNet2DConditionModel(text_len=77, latent_repeat=3, **SMALL)
    u.load_state_dict(u.synthetic_state_dict(11))
    enc = CLIPTextModel(dtype=torch.float16, **CLIP_CFG)
    enc.load_state_dict(enc.synthetic_state_dict(31))
    vae = AutoencoderKL(block_out_channels=(64, 128), layers_per_block=1, sample_size=32)
    vae.load_state_dict(vae.synthetic_state_dict(5))
    tok = FakeCLIPTokenizer(vocab_size=CLIP_CFG["vocab_size"])
    with pytest.raises(ValueError):
        driver.run_job(args, SafeDenoiserPipeline(u, DDPMScheduler(), variant="threshold_time", vae=vae, text_encoder=enc, tokenizer=tok))
    pipe = SafeDenoiserPipeline(u, DDPMScheduler(), variant=driver.ERASE_IDS["sld"][1], vae=vae, text_encoder=enc, tokenizer=tok)
    E3, _, _ = pipe._new_encode_prompt(["a", "b"], None, safety_concept=pipe.safety_concept)
    assert E3.shape == (6, 77, 768) and torch.equal(E3[4], E3[5]) and not torch.equal(E3[2], E3[4])
    driver.run_job(args, pipe, prompts_per_batch=2)
    assert sorted(os.listdir(os.path.join(args.save_dir, "all"))) == ["21_sexual.png", "296_shocking.png"]
    assert pipe.last_stats["branches"] == 3 and pipe.last_stats["prompts"] == 2
    with pytest.raises(Exception):                                       # caller-supplied rows that do not match the prompt count
        pipe(["a", "b"], prompt_embeddings=E3[:4], num_inference_steps=2, return_latents=True, **driver.SLD_CONFIGS["MEDIUM"])


@pytest.mark.parametrize("config", ["copro_1k_config3", "coco_10k_config5"])
def test_baseline_config_3_and_5_tables_through_run_job_on_one_rank_of_two(tmp_path, config):
    """BASELINE configs 3 (CoPro_balanced_1k.csv: `idx, unsafe_prompt, safe_prompt, concept, category`; run_copro.py:436-448, the
    `fast` repellency module, :52) and 5 (datasets/coco_30k.csv: `case_number, source, prompt, evaluation_seed, coco_id`; run_coco30k.py:400-426) are
    prompt-sharded 8-GPU jobs; no multi-GPU node exists for this build, so what CAN run does: the table in its own dialect through
    `driver.run_job` as rank 1 of a 2-rank world on the engine (rows 1, 3, ... of the table, a per-rank tree, global case numbers)."""
    from safe_denoiser_amd.repellency import repellency_methods_fast as fast
    if config == "copro_1k_config3":
        rows = ["idx,unsafe_prompt,safe_prompt,concept,category"] + [
            f'{28731 + i},"An unsafe prompt number {i} about a quarrel","A safe prompt {i}",ostracism,Harrasment' for i in range(5)]
        erase_id, mod, cat_flag, want = "safree_neg_prompt_rep_time", fast, "nudity", ["28732_n-u-d-i-t-y.png", "28734_n-u-d-i-t-y.png"]
        params = {"scale": 0.03}
    else:
        # the header of the reference's own datasets/coco_30k.csv (the `prompt` + `case_number` dialect, no categories column)
        rows = ["case_number,source,prompt,evaluation_seed,coco_id"] + [f'{i},coco-30k,"A bicycle replica with a clock as wheel {i}.",{41337 + i},{203564 + i}'
                                                                        for i in range(5)]
        erase_id, mod, cat_flag, want = "safree_neg_prompt_rep_threshold_time", thr, "all", ["1_n-u-d-i-t-y.png", "3_n-u-d-i-t-y.png"]
        params = {"scale": 0.33, "sigma": 3.15, "beta_threshold": 1e-6, "beta_threshold_margin": 1e9}
    (tmp_path / "table.csv").write_text("\n".join(rows) + "\n")
    refs = orp.channel_normalise(torch.randn(12, 4, 16, 16, generator=torch.Generator().manual_seed(4)))
    torch.save(refs, tmp_path / "proj_ref.pt")
    task = {"repellency": {"method": "kernel_fast", "n_embed": 4, "guidance_scale": 0.0,
                           "params": dict(params, proj_ref_path=str(tmp_path / "proj_ref.pt"), cache_proj_ref=True)},
            "data": {"name": config}, "mean_processor": {}}
    (tmp_path / "task.yaml").write_text(yaml.dump(task))
    cfg = {"erase_id": erase_id, "safree": True, "svf": True, "lra": True, "nudity": cat_flag, "data": str(tmp_path / "table.csv"),
           "save_dir": str(tmp_path / "out"), "num_inference_steps": 6, "image_length": 128, "task_config": str(tmp_path / "task.yaml")}
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    args = driver.parse_args(["--config", str(tmp_path / "cfg.json")])
    tc = driver.load_task_config(args.task_config)
    u = UNet2DConditionModel(text_len=77, latent_repeat=3, **SMALL); u.load_state_dict(u.synthetic_state_dict(11))
    enc = CLIPTextModel(dtype=torch.float16, **CLIP_CFG); enc.load_state_dict(enc.synthetic_state_dict(31))
    vae = AutoencoderKL(block_out_channels=(64, 128), layers_per_block=1, sample_size=32); vae.load_state_dict(vae.synthetic_state_dict(5))
    sch = DDPMScheduler()
    pipe = SafeDenoiserPipeline(u, sch, variant=driver.ERASE_IDS[erase_id][1], vae=vae, text_encoder=enc,
                                tokenizer=FakeCLIPTokenizer(vocab_size=CLIP_CFG["vocab_size"]))
    kw = driver.repellency_kwargs(tc, args.num_inference_steps, sch)
    proc = mod.get_repellency_method(kw.pop("name"), torch.zeros(1, device="cuda"), None, None, **kw)
    t = {}
    art = driver.run_job(args, pipe, proc, tc, eval_func=lambda imgs, threshold: (False, 0.1), prompts_per_batch=2, rank=1, world=2, timings=t)
    assert art.save_dir.endswith("rank01") and [b["prompts"] for b in t["batches"]] == [2]          # rows 1 and 3 of five
    assert sorted(os.listdir(os.path.join(art.save_dir, "all"))) == want
    assert pipe.last_stats["branches"] == 3 and pipe.last_stats["window_steps"] > 0 and pipe.last_stats["window_readbacks"] == 0
    d = json.load(open(os.path.join(art.save_dir, "detect_dict.json")))
    assert d["unsafe"] == [False, False] and d["toxic_size"]["average"] == 2
