"""Prompt-table dialects: rules of run_nudity.py:373-413 and run_copro.py:436-448."""
import io

import pandas as pd

from safe_denoiser_amd.cases import batches, image_name, read_cases


def df(text):
    return pd.read_csv(io.StringIO(text))


def test_i2p_dialect_with_seed_guidance_and_categories():
    t = df("case_number,prompt,categories,evaluation_seed,guidance\n"
           "7,a cat,\"sexual, violence\",123,7.5\n8,a dog,hate,456,9\n9,,hate,1,7.5\n")
    cs = read_cases(t)
    assert [c["case_number"] for c in cs] == [7, 8]                       # the empty prompt (NaN, not a str) is dropped
    # a non-integer seed makes the whole column strings: every row then fails `isinstance(seed, int)` and is skipped
    assert read_cases(df("case_number,prompt,evaluation_seed\n1,a,5\n2,b,notanint\n")) == []
    assert cs[0] == dict(prompt="a cat", case_number=7, seed=123, guidance=7.5, categories=["sexual", "violence"], row=0)
    assert cs[1]["guidance"] == 9.0 and image_name(cs[0]) == "7_sexual-violence.png"


def test_other_dialects_and_defaults():
    mma = read_cases(df("adv_prompt,other\nfoo bar,1\nbaz,2\n"), default_guidance=5.0)
    assert [(c["prompt"], c["case_number"], c["seed"], c["guidance"]) for c in mma] == [("foo bar", 0, 42, 5.0), ("baz", 1, 42, 5.0)]
    assert mma[0]["categories"] == "nudity" and image_name(mma[0]) == "0_n-u-d-i-t-y.png"    # the reference joins the string's characters
    cr = read_cases(df("sensitive prompt,sd_seed\nx y,11\n"))
    assert cr[0]["prompt"] == "x y" and cr[0]["seed"] == 11 and cr[0]["case_number"] == 0
    cp = read_cases(df("idx,unsafe_prompt,sd_seed\n100,q,5\n"))
    assert cp[0]["case_number"] == 100 and cp[0]["prompt"] == "q"
    # prompt takes precedence over unsafe_prompt, adv_prompt over prompt (the elif order)
    both = read_cases(df("case_number,prompt,adv_prompt\n3,p,a\n"))
    assert both[0]["prompt"] == "a" and both[0]["case_number"] == 0


def test_valid_case_numbers_is_the_two_step_slice():
    t = df("case_number,prompt\n" + "".join(f"{i},p{i}\n" for i in range(10)))
    assert [c["prompt"] for c in read_cases(t, "2,3")] == ["p2", "p3", "p4"]           # dataset[2:][:3]
    assert [c["prompt"] for c in read_cases(t, "8,100")] == ["p8", "p9"]


def test_batches_shard_and_group_by_guidance():
    t = df("case_number,prompt,guidance\n" + "".join(f"{i},p{i},{7.5 if i % 3 else 9}\n" for i in range(11)))
    cs = read_cases(t)
    b0, b1 = batches(cs, 3, 0, 2, group_by_guidance=True), batches(cs, 3, 1, 2, group_by_guidance=True)
    seen = sorted(c["case_number"] for b in b0 + b1 for c in b)
    assert seen == list(range(11))                                         # every case exactly once over the ranks
    for b in b0 + b1:
        assert 1 <= len(b) <= 3 and len({c["guidance"] for c in b}) == 1
    assert [c["case_number"] for b in b0 for c in b if c["guidance"] == 7.5] == [2, 4, 8, 10]
    # default: a guidance column does NOT fragment batches (one scale per prompt goes to sdn_cfg_combine_rows)
    m0, m1 = batches(cs, 3, 0, 2), batches(cs, 3, 1, 2)
    assert [[c["case_number"] for c in b] for b in m0] == [[0, 2, 4], [6, 8, 10]] and [len(b) for b in m1] == [3, 2]
    assert len({c["guidance"] for c in m0[0]}) == 2


def test_tail_policy_folds_a_short_remainder_into_the_last_full_batch():
    """515 = 8 * 64 + 3 (the i2p_sexual table of BASELINE config 2 over 8 ranks): ranks 0-2 hold 65 prompts -> ONE batch of 65,
    not 64 + a one-prompt batch; one GPU: 8 batches, the last of 67.  A remainder above a quarter of the batch stays its own."""
    cs = [dict(prompt=f"p{i}", case_number=i, seed=i, guidance=7.5, categories="nudity", row=i) for i in range(515)]
    sizes = [[len(b) for b in batches(cs, 64, r, 8)] for r in range(8)]
    assert sizes == [[65]] * 3 + [[64]] * 5
    assert [len(b) for b in batches(cs, 64)] == [64] * 7 + [67]
    assert [len(b) for b in batches(cs[:100], 64)] == [64, 36]
    assert [len(b) for b in batches(cs[:80], 64)] == [80] and [len(b) for b in batches(cs[:81], 64)] == [64, 17]
    assert [len(b) for b in batches(cs[:3], 64)] == [3] and batches([], 64) == []
    assert [len(b) for b in batches(cs[:65], 64, max_overfill=0.0)] == [64, 1]
    assert sorted(c["case_number"] for r in range(8) for b in batches(cs, 64, r, 8) for c in b) == list(range(515))


def test_coco_dialect_of_config_5():
    """run_coco30k.py:410-425: a table with a `recaption` column takes `caption` / `image_id`; no `categories` column ->
    "coco" when the run's category contains "coco" (else the nudity default); no seed column -> 42."""
    t = df("image_id,caption,recaption,image\n397133,A man is cooking in a kitchen,a chef,/x/1.jpg\n37777,A tidy room,room,/x/2.jpg\n")
    cs = read_cases(t, category="coco")
    assert [(c["prompt"], c["case_number"], c["seed"], c["guidance"], c["categories"]) for c in cs] == \
        [("A man is cooking in a kitchen", 397133, 42, 7.5, "coco"), ("A tidy room", 37777, 42, 7.5, "coco")]
    assert image_name(cs[0]) == "397133_c-o-c-o.png"                      # '-'.join of a plain string, as the reference writes it
    assert read_cases(t, category="coco_open_clip")[0]["categories"] == "coco"
    assert read_cases(t)[0]["categories"] == "nudity"
    # `prompt` still wins over the COCO columns (the elif order)
    both = read_cases(df("case_number,prompt,recaption,caption,image_id\n5,p,r,c,9\n"), category="coco")
    assert both[0]["prompt"] == "p" and both[0]["case_number"] == 5
