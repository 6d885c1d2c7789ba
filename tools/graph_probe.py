"""Developer probe: the train_rqvae step flow (k-means warm-up, eager steps, hipGraph capture, mixed eager/graph
replays incl. epoch-tail batches) outside train(); works stand-alone, see the note in train_rqvae.py."""
import os, sys, faulthandler, warnings
faulthandler.enable(); faulthandler.dump_traceback_later(30, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rq-vae-recommender_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from data.schemas import SeqBatch
from data.processed import ItemData
from modules.quantize import QuantizeForwardMode
from modules.rqvae import RqVae
from modules.tokenizer.semids import SemanticIdTokenizer
from rqhip import tuning, dist as rqdist
import train_rqvae
variant = sys.argv[1]
tuning.enable_tuned_gemms()
torch.manual_seed(0); np.random.seed(0)
B = 640
dev = torch.device("cuda", 0)
ds = ItemData(root="synthetic:3000", train_test_split="train").to_device(dev)
batches = train_rqvae._DeviceBatcher(ds, B)
m = RqVae(input_dim=768, embed_dim=32, hidden_dims=[512, 256, 128], codebook_size=256, n_layers=3, n_cat_features=0,
          codebook_kmeans_init=True, codebook_mode=QuantizeForwardMode.STE).to(dev)
opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-4, fused=True, capturable=True)
red = rqdist.FlatGradReducer(m.parameters())
if "tok" in variant:
    tok = SemanticIdTokenizer(input_dim=768, hidden_dims=[512, 256, 128], output_dim=32, codebook_size=256, n_layers=3, n_cat_feats=0)
    tok.rq_vae = m
g = train_rqvae._GraphedStep(m, opt, red, B, 768, dev, 0.2)
for it in range(12):
    m.train()
    if it == 0:
        m(ds[torch.arange(min(20000, len(ds)))], 0.2)
    data = next(batches)
    if it >= 3 and data.x.shape[0] == B:
        if g.graph is None:
            g.capture(); print("captured", flush=True)
        out = g.run(data.x)
    else:
        red.zero_()
        out = m(data, gumbel_t=0.2)
        loss = out.loss / 1
        loss.backward()
        tl = loss.detach()
        out = type(out)(*[v.detach() for v in out])
        del loss
        red.allreduce_mean(); opt.step()
    print(it, data.x.shape[0], float(out.loss.detach()), flush=True)
print(variant, "ok")
