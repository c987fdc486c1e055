"""Prompt-table dialects of the reference's drivers (the data format on the input side of the loop).

`read_cases` turns a CSV / DataFrame into the per-prompt records the drivers build inline:
run_nudity.py:373-408 (`valid_case_numbers` slicing; `adv_prompt` (MMA-diffusion), `sensitive prompt` (concept removal),
`prompt` + `case_number` (i2p / RECE tables); `guidance` column or the CLI default; `evaluation_seed` else `sd_seed` else
42; `categories` split on ", " else "nudity"; rows whose prompt is not a string or whose seed is not an int are skipped,
:411-413), run_copro.py:436-448 (`unsafe_prompt` + `idx`) and run_coco30k.py:410-425 (the COCO-30k table of BASELINE
config 5: a row that HAS a `recaption` column takes its prompt from `caption` and its case number from `image_id`; rows
without a `categories` column are labelled "coco" when the run's --category contains "coco").  `batches` groups them for the
batched engine loop: each prompt keeps its own seed -> generator and its own guidance scale (sdn_cfg_combine_rows takes one
scale per prompt, so a `guidance` column does not fragment batches), and a short remainder is folded into the last full batch.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

# (column whose PRESENCE selects the dialect, prompt column, case-number column or None = the row index), in the reference's
# if / elif order (run_nudity.py:379-388, run_coco30k.py:400-413, run_copro.py:442-445)
_PROMPT_COLUMNS = (("adv_prompt", "adv_prompt", None), ("sensitive prompt", "sensitive prompt", None),
                   ("prompt", "prompt", "case_number"), ("recaption", "caption", "image_id"),
                   ("unsafe_prompt", "unsafe_prompt", "idx"))


def _is_int(v) -> bool:
    import numbers
    return isinstance(v, numbers.Integral) and not isinstance(v, bool)


def read_cases(table, valid_case_numbers: str = "0,100000", default_guidance: float = 7.5, category: str = "nudity") -> List[dict]:
    """`table`: path to a CSV or a pandas DataFrame; `category`: the run's --category (only "coco" in it matters, see above).
    Returns [{prompt, case_number, seed, guidance, categories, row}]."""
    import pandas as pd
    df = pd.read_csv(table) if isinstance(table, (str, bytes)) or hasattr(table, "__fspath__") else table
    vstart, vend = (int(x) for x in valid_case_numbers.split(","))
    df = df[vstart:][:vend]                                        # the reference's two-step slice (:373-375)
    out = []
    for it, data in df.iterrows():
        prompt = case = None
        for key_col, col, case_col in _PROMPT_COLUMNS:
            if key_col in data:
                prompt = data[col]
                case = it if case_col is None else data[case_col]
                break
        if prompt is None:
            continue
        guidance = data["guidance"] if "guidance" in data else default_guidance
        if "evaluation_seed" in data:
            seed = data["evaluation_seed"]
        elif "sd_seed" in data:
            seed = data["sd_seed"]
        else:
            seed = 42
        if "categories" in data and isinstance(data["categories"], str):
            cats = data["categories"].split(", ")
        else:
            cats = "coco" if "coco" in category else "nudity"        # run_coco30k.py:421-426
        if hasattr(seed, "item"):
            seed = seed.item()
        if hasattr(guidance, "item"):
            guidance = guidance.item()
        if hasattr(case, "item"):
            case = case.item()
        if not isinstance(prompt, str) or not _is_int(seed) or not isinstance(guidance, (int, float)):
            continue                                               # "check if data is broken" (:411-413)
        out.append(dict(prompt=prompt, case_number=case, seed=int(seed), guidance=float(guidance), categories=cats, row=it))
    return out


def image_name(case: dict) -> str:
    """File name of the drivers' outputs: `{case_num}_{'-'.join(categories)}.png` (run_nudity.py:485-504)."""
    cats = case["categories"]
    return f"{case['case_number']}_{'-'.join(cats)}.png"           # a plain string joins its characters, as the reference does


def batches(cases: Iterable[dict], prompts_per_batch: int, rank: int = 0, world: int = 1, group_by_guidance: bool = False,
            max_overfill: float = 0.25) -> List[List[dict]]:
    """This rank's cases (`rank::world`, as dist.shard_indices) cut into batches of `prompts_per_batch` prompts in table order.
    Tail policy: a remainder of at most `max_overfill * prompts_per_batch` prompts joins the last full batch instead of
    running as a batch of its own -- 515 prompts over 8 ranks at 64 per batch give ranks 0-2 ONE batch of 65, not 64 + 1 (a
    one-prompt batch costs a new launch plan and a whole 50-step loop at a fraction of the machine).  A longer remainder stays
    a batch of its own.  `group_by_guidance=True` restores batches that share one guidance scale (for a pipeline that takes
    a scalar only)."""
    mine = list(cases)[rank::world]
    if group_by_guidance:
        groups: dict = {}
        for c in mine:
            groups.setdefault(c["guidance"], []).append(c)
        streams = list(groups.values())
    else:
        streams = [mine]
    P = int(prompts_per_batch)
    if P <= 0:
        raise ValueError("prompts_per_batch must be positive")
    out = []
    for g in streams:
        full, rem = divmod(len(g), P)
        cuts = [P] * full
        if rem:
            if full and rem <= max_overfill * P:
                cuts[-1] += rem
            else:
                cuts.append(rem)
        lo = 0
        for n in cuts:
            out.append(g[lo:lo + n])
            lo += n
    return out


def generators(batch: List[dict], device="cuda"):
    """One torch.Generator per prompt, seeded like `gen.manual_seed(seed)` (run_nudity.py:448)."""
    import torch
    return [torch.Generator(device=device).manual_seed(c["seed"]) for c in batch]
