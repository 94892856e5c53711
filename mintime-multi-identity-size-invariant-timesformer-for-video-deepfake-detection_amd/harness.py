"""The reference callers' step bodies (train.py:332-381, test.py:235-247) on the MI355X modules, for synthetic
inputs.  train.py / test.py themselves import tensorflow, cv2, albumentations, timm ... and cannot run offline;
this reproduces exactly what they do with the two models (same rearranges, same kwargs, same loss, same optimizer)."""
import torch
import torch.nn.functional as F

from . import arch, optim, synth
from .efficientnet import EfficientNet
from .timesformer import SizeInvariantTimeSformer


def build_models(num_frames=8, seed=0, device="cuda", require_attention=False, drop_connect_rate=arch.DROP_CONNECT_RATE,
                 train_extractor=True):
    cfg = arch.default_tsf_config(1280, num_frames)
    ef = EfficientNet.from_name("efficientnet-b0", drop_connect_rate=drop_connect_rate)
    ef.load_state_dict(synth.effnet_b0_state(seed))
    tsf = SizeInvariantTimeSformer(config=cfg, require_attention=require_attention)
    tsf.load_state_dict(synth.tsf_state(cfg, seed))
    ef.to(device).train(train_extractor)
    tsf.to(device).train()
    return cfg, ef, tsf


def build_models_xs(num_frames=16, seed=0, device="cuda", require_attention=False):
    """BASELINE config 5 (the "XS" variant): Xception extractor (models/xception.py) + TimeSformer over 2048-channel features."""
    from .xception import xception
    cfg = arch.default_tsf_config(2048, num_frames)
    xc = xception(num_classes=1, pretrain_path=None)
    xc.load_state_dict(synth.xception_state(seed))
    tsf = SizeInvariantTimeSformer(config=cfg, require_attention=require_attention)
    tsf.load_state_dict(synth.tsf_state(cfg, seed))
    xc.to(device).train()
    tsf.to(device).train()
    return cfg, xc, tsf


def make_optimizer(cfg, ef, tsf):
    """train.py:180-190: optimizer over chain(extractor, model) parameters; SGD(lr, weight_decay) from the YAML."""
    t = cfg["training"]
    params = list(ef.parameters()) + list(tsf.parameters())
    kind = str(t.get("optimizer", "sgd")).lower()
    fused = {"sgd": optim.FusedSGD, "adamw": optim.FusedAdamW, "adam": optim.FusedAdam}
    plain = {"sgd": torch.optim.SGD, "adamw": torch.optim.AdamW, "adam": torch.optim.Adam}
    if kind not in fused:
        raise ValueError("Error: Invalid optimizer specified in the config file.")      # train.py:191-193
    # one multi-tensor launch per step instead of torch's foreach kernels (same update rule)
    return (fused if params[0].is_cuda else plain)[kind](params, lr=t["lr"], weight_decay=t["weight-decay"])


def device_batch(batch, num_frames=8, num_identities=2, seed=0, device="cuda", ragged=False, as_uint8=False):
    """Synthetic clips resident on the device: videos are generated there (uint8-valued fp32, BGR 0..255; the reference's
    dataset hands over fp32, deepfakes_dataset.py:339 -- `as_uint8` keeps the raw bytes, which the stems also accept)."""
    aux = synth.clip_inputs(batch, num_frames, num_identities, seed, ragged=ragged, with_video=False)
    g = torch.Generator(device=device).manual_seed(1000 + seed)
    videos = torch.randint(0, 256, (batch, num_frames, 224, 224, 3), generator=g, device=device, dtype=torch.uint8)
    if ragged:
        videos = videos * aux["mask"].to(device)[:, :, None, None, None].to(torch.uint8)
    if not as_uint8:
        videos = videos.float()
    return dict(videos=videos, mask=aux["mask"].to(device), identities_mask=aux["identities_mask"].to(device),
                size_embedding=aux["size_embedding"], positions=aux["positions"].to(device),
                labels=aux["labels"].to(device))


def forward(ef, tsf, batch):
    videos = batch["videos"]
    b, f, h, w, c = videos.shape
    x = videos.reshape(b * f, h, w, c).permute(0, 3, 1, 2)                    # train.py:341 (a view)
    features = ef(x)                                                           # train.py:348
    features = features.reshape(b, f, *features.shape[1:])                    # train.py:354 (a view)
    return tsf(features, mask=batch["mask"], size_embedding=batch["size_embedding"],
               identities_mask=batch["identities_mask"], positions=batch["positions"])   # train.py:355


def train_step(ef, tsf, optimizer, batch, reducer=None, pos_weight=None):
    """One optimisation step (train.py:367-378).  The loss stays on the device (the reference moves the logits to the
    CPU first, one D2H sync per step, train.py:367)."""
    y_pred = forward(ef, tsf, batch)
    if isinstance(y_pred, tuple):
        y_pred = y_pred[0]
    loss = optim.bce_with_logits(y_pred, batch["labels"], pos_weight)          # value + gradient in one launch, on the device
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    if reducer is not None:
        reducer.allreduce()
    optimizer.step()
    return loss


@torch.no_grad()
def eval_step(ef, tsf, batch):
    """test.py:235-247 / predict.py:401-406: eval forward; returns logits (and attentions if the model was built with them)."""
    return forward(ef, tsf, batch)


def aggregate_attentions(attentions, heads, num_frames, frames_per_identity, scale_factor=50000):
    """Reference utils.py:68-96 on the device: attentions = [space, time] as returned by the model with
    require_attention=True.  Returns (aggregated [space, time, combined] lists of num_frames floats, per-identity sums) with the
    reference's own slicing rule for the identity sums (utils.py:87-94)."""
    from . import lib as L
    space, time_ = attentions
    bh, _, n = space.shape
    out = torch.empty(3, num_frames, dtype=torch.float32, device=space.device)
    L.check(L.get().mt_attn_aggregate(L.ptr(space.contiguous()), L.ptr(time_.contiguous()), L.ptr(out), bh, n, num_frames,
                                      float(scale_factor), L.stream_ptr()), "mt_attn_aggregate")
    agg = out.cpu().tolist()
    comb = agg[-1]
    identity = []
    for index, frames in enumerate(frames_per_identity):
        if index == 0:
            identity.append(sum(comb[:frames - 1]))
        else:
            identity.append(sum(comb[frames_per_identity[index - 1] - 1:frames - 1]))
    return agg, identity


class GraphedEval:
    """Eval forward captured once into a HIP graph (torch.cuda.CUDAGraph) and replayed: the ~440 kernel launches of
    EfficientNet-B0 + TimeSformer become one graph launch, which removes the per-launch host cost that dominates small-batch
    inference (test.py runs bs = 1).  Inputs are copied into static buffers; outputs are static tensors overwritten by replay."""

    def __init__(self, ef, tsf, example_batch):
        assert not ef.training and not tsf.training, "capture the eval() forward"
        self.ef, self.tsf = ef, tsf
        dev = next(tsf.parameters()).device
        # every input lives in a static device buffer (an H2D copy of the reference's host-side size_embedding is not capturable)
        self.static = {k: (v.to(dev).clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(2):                      # warm-up: loads kernels, sets LDS attributes, fills the allocator
                forward(ef, tsf, self.static)
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.out = forward(ef, tsf, self.static)

    def __call__(self, batch):
        for k, v in batch.items():
            if torch.is_tensor(v):
                self.static[k].copy_(v, non_blocking=True)
        self.graph.replay()
        return self.out
