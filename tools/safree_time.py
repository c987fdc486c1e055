import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from safe_denoiser_amd.clip import CLIPTextModel
from safe_denoiser_amd.pipeline import SafeDenoiserPipeline, make_scheduler
from tests_support.fake_tokenizer import FakeCLIPTokenizer
dev = torch.device("cuda", 0)
enc = CLIPTextModel(); enc.load_synthetic_on_device(4242, device=dev)
class U:  # placeholder: _safree_prepare does not touch the UNet
    pass
pipe = SafeDenoiserPipeline(U(), make_scheduler("ddpm"), text_encoder=enc, tokenizer=FakeCLIPTokenizer())
prompts = [bench.synthetic_prompt(i) for i in range(64)]
E, _ids, am = pipe._new_encode_prompt(prompts, ", ".join(bench.NEG_SPACE))
res = {}
for batched in (True, False, True, False):
    pipe.batched_safree = batched
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = pipe._safree_prepare(prompts, E, am, bench.NEG_SPACE, dict(bench.SAFREE))
    torch.cuda.synchronize()
    print(f"batched={batched}: {(time.perf_counter() - t0) * 1e3:.1f} ms; removed {sum(r['n_removed'])}, steps {sum(r['beta_adjusted'])}")
    res[batched] = r
print("same decisions:", res[True]["n_removed"] == res[False]["n_removed"], res[True]["beta_adjusted"] == res[False]["beta_adjusted"],
      "max |d emb|", float((res[True]["rescaled_text_embeddings"] - res[False]["rescaled_text_embeddings"]).abs().max()))
