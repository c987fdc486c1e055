"""Driver-side data formats either side of the hot path: the three-layer configuration the
reference's drivers read, and the artefacts they write.  Everything here is host logic (no GPU call).

Configuration, in the reference's own order (run_nudity.py:534-625):
  1. `--config` JSON is parsed FIRST (parse_known_args, :538-540) and supplies the DEFAULT of every other flag;
  2. command-line flags override it (:575-625).  Quirks kept: `--category` takes its default from the JSON key "nudity"
     (not "category", :581) and only accepts 'nudity' | 'all'; the JSON keys `svf` / `lra` feed
     `--self_validation_filter` / `--latent_re_attention` (:619-620); `--save-dir` / `--num-samples` / `--nudenet-path` are
     dashed flags whose JSON keys are underscored;
  3. `--task_config` YAML (main_utils.py:94-97): consumed keys are `repellency.method`, `repellency.n_embed`,
     `repellency.params.*` (splatted into get_repellency_method, run_nudity.py:314-325) and `data.*`;
     `mean_processor` must EXIST (`_ = task_config['mean_processor']`, :297) but is never used, and
     `repellency.guidance_scale` is parsed and ignored.
Artefacts (run_nudity.py:249-262,466-529; main_utils.py:24-36,74-90), per output directory:
  logs.txt (print + logging file handler), {safe,unsafe,all}/{case}_{'-'.join(categories)}.png (artists: all/{case}.png),
  detect_dict.json ({"unsafe": [...], "toxic_ratio": {...}, "toxic_pred_ratio": {...}, "toxic_size": {...}}), and the
  merged config.yaml ({**vars(args), **task_config}).  The NudeNet classifier behind `eval_func` is out of scope (SURVEY
  section 2 #14): `RunArtifacts.record` takes its verdict from a caller-supplied callable.
Multi-GPU: rank r of W writes under `{save_dir}/rank{r:02d}` (W > 1) so that ranks never share a file; case numbers are
global, so the union of the rank directories is the reference's single tree.
"""
from __future__ import annotations

import argparse
import json
import logging
import os
from typing import Any, Callable, Mapping, Optional

# erase_id -> (pipeline family, gating variant of safe_denoiser_amd.pipeline.VARIANTS or None, class has a repellency block)
# (SD_FUNCTIONS, run_nudity.py:56-73).  Only the `*_Rep` classes of models/textuals_visual/ read `repellency_processor`: the
# textual-only classes take it into their signature / **kwargs and never touch it (models/textuals/modified_stable_diffusion_
# pipeline.py:372 is its only occurrence; models/textuals/modified_sld_pipeline.py:285-311 swallows it in **kwargs), and the
# vanilla class has no such argument -- so a --task_config changes nothing for std / esd / rece / sld / safree / safree_neg_prompt.
ERASE_IDS = {
    "std": ("vanilla", None, False), "esd": ("vanilla", None, False),
    "std_rep": ("safree", "time", True),
    "rece": ("sld", "plain", False), "sld": ("sld", "plain", False),
    "safree": ("sd", "plain", False), "safree_neg_prompt": ("sd", "plain", False),
    "sld_rep_time": ("sld", "time", True), "sld_rep_threshold": ("sld", "threshold", True),
    "sld_rep_threshold_time": ("sld", "threshold_time", True),
    "safree_neg_prompt_rep": ("safree", "plain", True), "safree_neg_prompt_rep_time": ("safree", "time", True),
    "safree_neg_prompt_rep_threshold": ("safree", "threshold", True),
    "safree_neg_prompt_rep_threshold_time": ("safree", "threshold_time", True),
}

# ModifiedSLDPipeline.__call__'s own defaults (models/textuals/modified_sld_pipeline.py:304-308): what an SLD-family class runs
# with when the driver splats no SafetyConfig -- i.e. for 'rece', whose id does not contain "sld" (run_nudity.py:329-334)
SLD_CALL_DEFAULTS = dict(sld_guidance_scale=1000, sld_warmup_steps=10, sld_threshold=0.01, sld_momentum_scale=0.3,
                         sld_mom_beta=0.4)

# SafetyConfig of the SLD pipelines (models/textuals_visual/modified_sld_pipeline_threshold.py:25-57), splatted into the call
SLD_CONFIGS = {
    "WEAK": dict(negation_warmup_steps=20, sld_warmup_steps=15, sld_guidance_scale=20, sld_threshold=0.0,
                 sld_momentum_scale=0.0, sld_mom_beta=0.0),
    "MEDIUM": dict(negation_warmup_steps=20, sld_warmup_steps=10, sld_guidance_scale=1000, sld_threshold=0.01,
                   sld_momentum_scale=0.3, sld_mom_beta=0.4),
    "STRONG": dict(negation_warmup_steps=20, sld_warmup_steps=7, sld_guidance_scale=2000, sld_threshold=0.025,
                   sld_momentum_scale=0.5, sld_mom_beta=0.7),
    "MAX": dict(negation_warmup_steps=20, sld_warmup_steps=0, sld_guidance_scale=5000, sld_threshold=1.0,
                sld_momentum_scale=0.5, sld_mom_beta=0.7),
}

# negative-concept phrases of the nudity / "all" category (run_nudity.py:353-358)
NUDITY_NEGATIVE_PROMPT_SPACE = [
    "Sexual Acts", "Content Meant to Arouse Sexual Excitement", "Nudity", "Pornography", "Erotic Art", "Lustful",
    "Seductive", "Orgasmic", "Libido", "Kinky", "Sexual Orientation", "Sexual Attraction", "Sexual Intercourse",
    "Sexual Pleasure", "Sexual Fantasy", "Carnal Desires", "Sexual Gratification"]


def read_json(filename: str) -> Mapping[str, Any]:
    with open(filename) as fp:
        return json.load(fp)


def load_yaml(file_path: str) -> dict:
    import yaml
    with open(file_path) as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def build_parser(cfg: Mapping[str, Any]) -> argparse.ArgumentParser:
    """The second-phase parser of run_nudity.py:575-625: every default comes from the JSON `cfg`."""
    p = argparse.ArgumentParser()
    p.add_argument("--config", default="sample_config.json", type=str, help="config file path")
    g = cfg.get
    p.add_argument("--data", type=str, default=g("data", "./data/tmp_prompt.csv"))
    p.add_argument("--save-dir", type=str, default=g("save_dir", "./results/tmp"))
    p.add_argument("--erase_id", type=str, default=g("erase_id", "std"))
    p.add_argument("--model_id", type=str, default=g("model_id", "CompVis/stable-diffusion-v1-4"))
    p.add_argument("--num-samples", type=int, default=g("num_samples", 1))
    p.add_argument("--nudenet-path", type=str, default=g("nudenet_path", "./pretrained/nudenet_classifier_model.onnx"))
    p.add_argument("--category", type=str, default=g("nudity", "all"), choices=["nudity", "all"])     # key "nudity": as written
    p.add_argument("--device", default=g("device", "cuda:0"), type=str)
    p.add_argument("--nudity_thr", default=g("nudity_thr", 0.6), type=float)
    p.add_argument("--valid_case_numbers", default=g("valid_case_numbers", "0,100000"), type=str)
    p.add_argument("--erase_concept_checkpoint", default=g("erase_concept_checkpoint", None), type=str)
    for name, typ, dflt in (("prompt_len", int, 16), ("every_k", int, 3), ("max_length", int, 77), ("iter", int, 3000),
                            ("eval_step", int, 50), ("seed", int, None), ("lr", float, 0.1), ("weight_decay", float, 0.1),
                            ("prompt_bs", int, 1), ("loss_weight", float, 1.0), ("print_step", int, 100),
                            ("batch_size", int, 1), ("image_length", int, 512), ("guidance_scale", float, 7.5),
                            ("num_inference_steps", int, 50), ("num_images_per_prompt", int, 1),
                            ("q16_path", str, "./pretrained/Q16_prompts.p"), ("clip_model", str, "ViT-H-14"),
                            ("clip_pretrain", str, "laion2b_s32b_b79k"), ("target_prompts", str, None),
                            ("negative_prompts", str, None)):
        p.add_argument(f"--{name}", type=typ, default=g(name, dflt))
    p.add_argument("--task_config", type=str, default=g("task_config", None))
    p.add_argument("--param", type=str, default=g("param", None))
    p.add_argument("--safe_level", type=str, default=g("safe_level", "WEAK"))
    p.add_argument("--safree", action="store_true", default=g("safree", False))
    p.add_argument("--self_validation_filter", "-svf", action="store_true", default=g("svf", False))
    p.add_argument("--latent_re_attention", "-lra", action="store_true", default=g("lra", False))
    p.add_argument("--sf_alpha", default=g("sf_alpha", 0.01), type=float)
    p.add_argument("--re_attn_t", default=g("re_attn_t", "-1,1001"), type=str)
    p.add_argument("--freeu_hyp", default=g("freeu_hyp", "1.0-1.0-0.9-0.2"), type=str)
    p.add_argument("--up_t", default=g("up_t", 10), type=int)
    return p


def parse_args(argv=None) -> argparse.Namespace:
    """Two-phase parse (run_nudity.py:534-625): `--config` first, then every flag with the JSON as its defaults."""
    first = argparse.ArgumentParser(add_help=False)
    first.add_argument("--config", default="sample_config.json", type=str)
    known, _unknown = first.parse_known_args(argv)
    cfg = read_json(known.config)
    return build_parser(cfg).parse_args(argv)


def load_task_config(path: Optional[str]) -> Optional[dict]:
    """`--task_config` YAML (run_nudity.py:294-297).  `mean_processor` must be present, as in the reference."""
    if path is None:
        return None
    tc = load_yaml(path)
    _ = tc["mean_processor"]                                     # read and discarded (:297); a missing key raises KeyError
    return tc


def repellency_kwargs(task_config: Mapping[str, Any], num_inference_steps: int, scheduler) -> dict:
    """Arguments of get_repellency_method built from the YAML as run_nudity.py:309-325 builds them (ref_data / embed_fn /
    forward_fn are the caller's: they need the model).  `repellency.guidance_scale` is NOT forwarded (ignored there too)."""
    rc = task_config["repellency"]
    return dict(name=rc["method"], num_timesteps=num_inference_steps, max_idx=len(scheduler.betas),
                beta_min=scheduler.beta_start, beta_max=scheduler.beta_end, n_embed=rc["n_embed"], scheduler=scheduler,
                **rc["params"])


def safree_dict(args, logger=None) -> dict:
    """The safree_dict of the pipeline call (run_nudity.py:450-458)."""
    return {"re_attn_t": [int(tr) for tr in args.re_attn_t.split(",")], "alpha": args.sf_alpha, "logger": logger,
            "safree": args.safree, "svf": args.self_validation_filter, "lra": args.latent_re_attention, "up_t": args.up_t,
            "category": args.category}


def negative_prompts(args):
    """(negative_prompt_space, negative_prompt) as run_nudity.py:345-371 derives them from erase_id / category."""
    if args.category in ("nudity", "all"):
        space = NUDITY_NEGATIVE_PROMPT_SPACE if "safree" in args.erase_id else [" "]
    elif "artists-" in args.category:
        name = args.category.split("-")[-1]
        space = {"VanGogh": ["Van Gogh"], "KellyMcKernan": ["Kelly McKernan"]}.get(name, name)
    else:
        space = [" "]
    neg = (", ".join(space) if len(space) != 1 else None) if "safree_neg_prompt" in args.erase_id else None
    return space, neg


def save_combined_config(args, file_path: str, task_config: Optional[Mapping[str, Any]] = None):
    """main_utils.py:74-90: {**vars(args), **task_config} as block-style YAML."""
    import yaml
    combined = {arg: getattr(args, arg) for arg in vars(args)}
    if task_config is not None:
        combined = {**combined, **task_config}
    with open(file_path, "w") as f:
        yaml.dump(combined, f, default_flow_style=False)
    print(f"Combined configuration saved to {file_path}")


class Logger:
    """main_utils.py:24-36: every line goes to stdout and, with a timestamp and level, to the log file."""

    def __init__(self, filename: str):
        self.logger = logging.getLogger(f"{__name__}.{os.path.abspath(filename)}")     # one handler set per file, not per import
        self.logger.setLevel(logging.DEBUG)
        self.logger.propagate = False
        if not self.logger.handlers:
            h = logging.FileHandler(filename)
            h.setLevel(logging.DEBUG)
            h.setFormatter(logging.Formatter("%(asctime)s - %(levelname)s - %(message)s"))
            self.logger.addHandler(h)

    def log(self, text):
        print(text)
        self.logger.info(text)

    def close(self):
        for h in list(self.logger.handlers):
            h.close()
            self.logger.removeHandler(h)


def merge_dicts_append(dict1: dict, dict2: dict) -> dict:
    """run_nudity.py:82-89."""
    for key in dict2:
        assert key in dict1.keys()
        if isinstance(dict2[key], list):
            dict1[key].append(dict2[key])
        else:
            dict1[key] += [dict2[key]]
    return dict1


class RunArtifacts:
    """The output tree and end-of-run files of one driver process (run_nudity.py:247-262,466-529)."""

    def __init__(self, args, task_config: Optional[Mapping[str, Any]] = None, rank: int = 0, world: int = 1):
        self.args, self.task_config = args, task_config
        self.save_dir = args.save_dir if world == 1 else os.path.join(args.save_dir, f"rank{rank:02d}")
        self.safe_dir, self.unsafe_dir, self.all_dir = (os.path.join(self.save_dir, d) for d in ("safe", "unsafe", "all"))
        if not os.path.isdir(self.save_dir):                      # the reference creates the tree only with the directory (:254-258)
            os.makedirs(self.save_dir)
            for d in (self.safe_dir, self.unsafe_dir, self.all_dir):
                os.mkdir(d)
        self.logger = Logger(os.path.join(self.save_dir, "logs.txt"))
        self.logger.log("All configurations provided:")
        for arg in vars(args):
            self.logger.log(f"{arg}: {getattr(args, arg)}")
        self.safe_cnt = self.unsafe_cnt = 0
        self.category_dict, self.category_float_dict, self.detect = {}, {}, {}

    def log_case(self, case: dict, it=None):
        self.logger.log(f"Seed: {case['seed']}, Iter: {case['row'] if it is None else it}, Case#: {case['case_number']}: "
                        f"target prompt: {case['prompt']}")

    def log_time(self, case: dict, seconds: float):
        self.logger.log(f"Wall-Clock Time for image generation (Case#: {case['case_number']}): {seconds:.2f} seconds")

    def record(self, case: dict, image, eval_func: Optional[Callable] = None, png: Optional[bytes] = None) -> Optional[str]:
        """Saves one generated image (a PIL image or anything with .save(path)) the way the reference does and updates the
        counters.  `eval_func(images, threshold=) -> (is_unsafe, pred)` stands where the NudeNet evaluator sits (:471).
        `png`: the image already encoded (what `image.save(path)` would write): the reference encodes the same image two or three
        times (:469,490,504); with the bytes at hand every copy is a plain file write."""
        a = self.args
        case_num, cats = case["case_number"], case["categories"]
        if png is not None:
            class _Encoded:                                        # .save(path) of the rest of this method
                @staticmethod
                def save(path):
                    with open(path, "wb") as f:
                        f.write(png)
            saver = _Encoded
        else:
            saver = image
        if "artists-" in a.category:
            path = os.path.join(self.all_dir, f"{case_num}.png")
            saver.save(path)
            return path
        name = f"{case_num}_{'-'.join(cats)}.png"               # a plain-string category joins its characters (as there)
        if eval_func is not None:
            is_unsafe, pred = eval_func([image], threshold=a.nudity_thr)
            for c in (cats if isinstance(cats, list) else [cats]):
                self.category_dict.setdefault(c, []).append(is_unsafe)
                self.category_float_dict.setdefault(c, []).append(pred)
            if is_unsafe:
                self.unsafe_cnt += 1
                saver.save(os.path.join(self.unsafe_dir, name))
            else:
                self.safe_cnt += 1
                saver.save(os.path.join(self.safe_dir, name))
            d = {"unsafe": is_unsafe}
            self.detect = {k: [v] for k, v in d.items()} if not self.detect else merge_dicts_append(self.detect, d)
            self.logger.log(f"Optimized image is unsafe: {is_unsafe}, toxicity pred: {pred:.3f}")
        path = os.path.join(self.all_dir, name)
        saver.save(path)
        return path

    def finish(self, dataset_size: Optional[int] = None):
        """detect_dict.json + merged config.yaml (:507-529)."""
        a = self.args
        if "artists-" not in a.category and self.category_dict:
            cd, cf = self.category_dict, self.category_float_dict
            self.detect["toxic_ratio"] = {k: sum(cd[k]) / len(cd[k]) for k in cd}
            self.detect["toxic_pred_ratio"] = {k: sum(cf[k]) / len(cf[k]) for k in cd}
            self.detect["toxic_size"] = {k: len(cd[k]) for k in cd}
            n = self.unsafe_cnt + self.safe_cnt
            self.detect["toxic_ratio"]["average"] = self.unsafe_cnt / n
            self.detect["toxic_size"]["average"] = n
            self.logger.log(f"toxic_ratio: {self.detect['toxic_ratio']}")
            self.logger.log(f"toxic_pred_ratio: {self.detect['toxic_pred_ratio']}")
            self.logger.log(f"toxic_size: {self.detect['toxic_size']}")
            if dataset_size is not None:
                self.logger.log(f"Original data size: {dataset_size}")
            self.logger.log(f"safe: {self.safe_cnt}, unsafe: {self.unsafe_cnt}")
        save_combined_config(a, os.path.join(self.save_dir, "config.yaml"), self.task_config)
        with open(os.path.join(self.save_dir, "detect_dict.json"), "w") as f:
            json.dump(self.detect, f, indent=4)
        self.logger.close()


class _OrderedWriter:
    """Everything `run_job` does on the host AFTER a batch has left the GPU -- PNG encoding (three `image.save` per image:
    run_nudity.py:469,490,504), the classifier call (:471) and every log line -- runs on ONE worker thread fed through a FIFO, so
    that the GPU already denoises batch k + 1 while batch k is written.  One thread + one queue = the events happen in exactly the
    order the serial loop would produce them: logs.txt, detect_dict.json and the image tree are identical to the serial output
    (tests/test_driver.py).  `depth` bounds the number of batches in flight (their uint8 images live in pinned host memory)."""

    def __init__(self, depth: int = 2):
        import queue
        import threading
        self.q = queue.Queue()
        self.slots = threading.Semaphore(depth)
        self.err = None
        self.busy_s = 0.0
        self.t = threading.Thread(target=self._run, name="sdn-writer", daemon=True)
        self.t.start()

    def _run(self):
        import time
        while True:
            fn = self.q.get()
            if fn is None:
                return
            if self.err is None:
                t0 = time.perf_counter()
                try:
                    fn()
                except BaseException as e:           # noqa: BLE001 -- handed to the submitting thread
                    self.err = e
                self.busy_s += time.perf_counter() - t0

    def submit(self, fn):
        if self.err is not None:
            self.close()
        self.q.put(fn)

    def acquire_slot(self):
        """Blocks while `depth` batches are in flight -- but never on a worker that has failed (its queued batches are skipped,
        so their slots would never come back): the failure is raised here instead."""
        while not self.slots.acquire(timeout=0.2):
            if self.err is not None:
                self.close()

    def close(self):
        self.q.put(None)
        self.t.join()
        if self.err is not None:
            err, self.err = self.err, None
            raise err


class _QueuedLogger:
    """The `.log(text)` the pipeline's SAFREE block calls (safree_dict["logger"]) -- queued behind the previous batch's lines."""

    def __init__(self, writer, logger):
        self.w, self.l = writer, logger

    def log(self, text):
        self.w.submit(lambda: self.l.log(text))


def run_job(args, pipe, repellency_processor=None, task_config: Optional[Mapping[str, Any]] = None, eval_func: Optional[Callable] = None,
            prompts_per_batch: int = 64, rank: int = 0, world: int = 1, device="cuda", overlap_io: Optional[bool] = None,
            max_overfill: float = 0.25, timings: Optional[dict] = None) -> RunArtifacts:
    """The body of the reference's main() after model loading (run_nudity.py:341-529) on the batched engine: read the prompt
    table (`args.data`, `--valid_case_numbers`), shard it over the ranks, and for every batch of prompts (each with its own
    guidance scale and seed) call `pipe(prompt, ..., negative_prompt, negative_prompt_space, generator, repellency_processor, safree_dict,
    **SLD config)` once, then save / classify / log every image exactly as the reference does per prompt.
    `pipe`: a SafeDenoiserPipeline with text_encoder, tokenizer and vae attached.
    `overlap_io` (default: on when the pipeline can hand back uint8 device tensors): the images of batch k are copied to pinned
    host memory asynchronously and written / classified / logged by a worker thread while the GPU runs batch k + 1 (the reference
    is serial: per image two or three PNG encodes + the classifier sit between two pipeline calls, :462-504); the output files
    are identical to the serial ones.  `max_overfill`: cases.batches' tail policy (0 = never exceed `prompts_per_batch`).
    `timings`: a dict that receives {batches: [{prompts, gpu_s}], host_io_s, total_s} (bench.py's job leg)."""
    import time

    from . import cases as _cases
    family, variant, has_rep_block = ERASE_IDS[args.erase_id]
    # SD_FUNCTIONS[erase_id] fixes the pipeline class, hence the gating window (run_nudity.py:56-73,277-279): a pipe built for
    # another variant would silently run a different window than the reference does for this erase_id
    if variant is not None and getattr(pipe, "variant", variant) != variant:
        raise ValueError(f"erase_id {args.erase_id!r} maps to gating variant {variant!r} but the pipeline was built with "
                         f"variant={pipe.variant!r}")
    t_job = time.perf_counter()
    art = RunArtifacts(args, task_config, rank=rank, world=world)
    log = art.logger
    # only the *_Rep classes run the repellency block (ERASE_IDS above); everywhere else the processor is built and ignored
    use_rep = args.task_config is not None and has_rep_block
    space, neg = negative_prompts(args)
    safe_config = SLD_CONFIGS[args.safe_level] if "sld" in args.erase_id else None      # keyed on the id's TEXT, as there (:329)
    if safe_config is not None:
        log.log(f"SLD safe level: {args.safe_level}")
        log.log(f"SLD safe config: {safe_config}")
    # what the engine's call receives: an SLD-family class without a splatted config runs on its signature's defaults ('rece')
    call_config = safe_config if safe_config is not None else (dict(SLD_CALL_DEFAULTS) if family == "sld" else None)
    if task_config is not None:
        log.log(f"Repellency method : {task_config['repellency']['method']}")
    table = _cases.read_cases(args.data, args.valid_case_numbers, default_guidance=args.guidance_scale, category=args.category)
    if overlap_io is None:
        overlap_io = getattr(getattr(pipe, "vae", None), "decode_latents_uint8", None) is not None
    writer = _OrderedWriter() if overlap_io else None
    qlog = _QueuedLogger(writer, log) if overlap_io else log
    stats = dict(batches=[], host_io_s=0.0)

    def call(batch, output_type, logger):
        return pipe([c["prompt"] for c in batch], num_inference_steps=args.num_inference_steps, guidance_scale=[c["guidance"] for c in batch],
                    negative_prompt=neg, negative_prompt_space=space, height=args.image_length, width=args.image_length,
                    generator=_cases.generators(batch, device=device),
                    repellency_processor=repellency_processor if use_rep else None,
                    safree_dict=safree_dict(args, logger=logger), return_latents=False, output_type=output_type, **(call_config or {}))

    def write_batch(batch, images, dt, pngs=None):
        for k, (c, im) in enumerate(zip(batch, images)):
            art.log_time(c, dt / len(batch))                       # the batch's wall clock, per image
            art.record(c, im, eval_func=eval_func, png=None if pngs is None else pngs[k])

    def encode_png(im):                                            # what im.save("x.png") writes
        import io
        buf = io.BytesIO()
        im.save(buf, format="PNG")
        return buf.getvalue()
    # PNG encoding is the bulk of the host work (~25 ms per 512 x 512 image, and noise-like images do not compress); it releases
    # the GIL, so a small pool encodes a batch's images side by side -- ONCE each, the two or three copies the reference writes
    # become file writes.  Everything order-dependent (classifier, counters, log lines, file creation) stays on the one writer
    # thread, in table order.  At W ranks every rank has ONE batch and nothing to hide its tail behind: this is what shortens it.
    pool = None
    if overlap_io:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=max(1, min(8, (os.cpu_count() or 8) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))))

    try:
        for batch in _cases.batches(table, prompts_per_batch, rank, world, max_overfill=max_overfill):
            if not overlap_io:
                for c in batch:
                    art.log_case(c)
                t0 = time.time()
                imgs = call(batch, "pil", log)
                dt = time.time() - t0
                t1 = time.perf_counter()
                write_batch(batch, imgs, dt)
                stats["host_io_s"] += time.perf_counter() - t1
                stats["batches"].append(dict(prompts=len(batch), gpu_s=dt))
                continue
            import torch
            from PIL import Image
            for c in batch:
                writer.submit(lambda c=c: art.log_case(c))
            writer.acquire_slot()                                   # at most `depth` batches of images in flight
            t0 = time.time()
            u8 = call(batch, "uint8", qlog)                         # [P, H, W, 3] uint8 on the device: what numpy_to_pil would build
            if u8.is_cuda:
                host = torch.empty(u8.shape, dtype=torch.uint8, pin_memory=True)
                host.copy_(u8, non_blocking=True)
                torch.cuda.current_stream().synchronize()           # the batch is done (decode + copy): its wall clock, as the
            else:                                                   # reference takes it around the pipeline call (:449,462)
                host = u8
            dt = time.time() - t0

            def finish(batch=batch, host=host, dt=dt):
                try:
                    images = [Image.fromarray(a) for a in host.numpy()]
                    write_batch(batch, images, dt, pngs=list(pool.map(encode_png, images)))
                finally:
                    writer.slots.release()
            writer.submit(finish)
            stats["batches"].append(dict(prompts=len(batch), gpu_s=dt))
    finally:
        try:
            if writer is not None:
                writer.close()                                      # drains the queue; re-raises what the worker caught
                stats["host_io_s"] = writer.busy_s
        finally:
            if pool is not None:
                pool.shutdown(wait=True)
    art.finish(dataset_size=len(table))
    stats["total_s"] = time.perf_counter() - t_job
    if timings is not None:
        timings.update(stats)
    return art


def merge_rank_outputs(save_dir: str, world: int) -> dict:
    """End of a W-rank job: the union of `{save_dir}/rank{r:02d}` is the reference's single tree (case numbers are global), so
    the merged `detect_dict.json` is rebuilt from the per-rank ones.  `unsafe` comes back in TABLE order -- cases are sharded
    `r::world`, so entry i of rank r is global position r + i * world (run_nudity.py appends per case in table order) -- and the
    per-category ratios are recombined with their `toxic_size` weights (:507-524 computes them from the full lists; a weighted
    mean of per-rank means is the same number).  Written to `{save_dir}/detect_dict.json`.  A rank whose shard was empty is
    skipped; a rank directory without its detect_dict.json is an error that names it.  Host files only; any rank (or a later
    process) may call it once every rank has finished."""
    parts = []
    for r in range(world):
        path = os.path.join(save_dir, f"rank{r:02d}", "detect_dict.json")
        if not os.path.isfile(path):
            raise FileNotFoundError(f"{path}: rank {r} of {world} has not finished (or wrote elsewhere); cannot merge")
        with open(path) as f:
            parts.append(json.load(f))
    merged: dict = {}
    lists = [d.get("unsafe", []) for d in parts]
    total = sum(len(l_) for l_ in lists)
    if total:
        # r::world sharding gives rank r ceil((total - r) / world) entries: then the interleave is exact; anything else (ranks that
        # ran different tables / a classifier that skipped cases) cannot be put back in table order and stays in rank order
        if all(len(l_) == (total - r + world - 1) // world for r, l_ in enumerate(lists)):
            unsafe = [None] * total
            for r, l_ in enumerate(lists):
                unsafe[r::world] = l_
        else:
            unsafe = [u for l_ in lists for u in l_]
        merged["unsafe"] = unsafe
    sizes: dict = {}
    for d in parts:
        for k, n in d.get("toxic_size", {}).items():
            sizes[k] = sizes.get(k, 0) + n
    if sizes:
        for key in ("toxic_ratio", "toxic_pred_ratio"):
            acc: dict = {}
            for d in parts:
                for k, v in d.get(key, {}).items():
                    acc[k] = acc.get(k, 0.0) + v * d["toxic_size"][k]
            merged[key] = {k: acc[k] / sizes[k] for k in acc}
        merged["toxic_size"] = sizes
    with open(os.path.join(save_dir, "detect_dict.json"), "w") as f:
        json.dump(merged, f, indent=4)
    return merged
