import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mintime_amd
from mintime_amd import arch, synth, EfficientNet, effnet_engine
from oracle import mintime_oracle as O
from tests.util import rel_err
seed, B, Fr = 1, 2, 8
ef = EfficientNet.from_name("efficientnet-b0", drop_connect_rate=0.0)
sd = synth.effnet_b0_state(seed); ef.load_state_dict(sd); ef.train(True).cuda()
inp = synth.clip_inputs(B, Fr, 2, seed, ragged=True)
v = inp["videos"]; x = v.reshape(B*Fr, 224, 224, 3).permute(0, 3, 1, 2)
taps = {}
with torch.no_grad():
    ref = O.effnet_b0_forward(sd, x, training=True, taps=taps)
    ref64 = O.effnet_b0_forward(O.to_dtype(sd, torch.float64), x.double(), training=True)
    feat, ys = effnet_engine.effnet_apply(ef, x.cuda(), want_blocks=True)
for i, y in enumerate(ys):
    print("block", i, "rel err vs oracle32 %.3e" % rel_err(y, taps[f"block{i}"]))
print("feat vs oracle32 %.3e  vs oracle64 %.3e ; oracle32 vs oracle64 %.3e" % (rel_err(feat, ref), rel_err(feat, ref64), rel_err(ref, ref64)))
