"""Generation harness: this repo's counterpart of the reference's `evaluate_per_rank` / `gather_generated_results`
(pointnet2/mesh_evaluation.py:15-186) for the latent-DDPM path.

Multi-GPU (one process per GPU, `torch.distributed`): samples are independent (GroupNorm only, no cross-sample op), so
the requested samples are split into contiguous per-rank slices exactly like the reference's per-rank datasets
(ceil division, last rank short: pointnet2/shapenet_psr_dataloader/npz_dataset.py:90-96) with NO data-path
collective; a single all-gather of the generated latents replaces the reference's per-rank npz files + rank-0 file
concatenation (mesh_evaluation.py:156-186).  Over xGMI this is RCCL (backend "nccl"); the CPU tests run it on gloo.
"""
import contextlib
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from .configs import CATEGORY_IDS, CATEGORY_NAMES


def init_distributed():
    """(rank, world, device, gather_device) of this process.  One process per GPU over RCCL (backend "nccl", bound to the rank's
    GPU); SLIDE_SHARE_GPU=1 (test knob for 1-GPU boxes): every rank uses device 0 and the collectives run on gloo / CPU tensors."""
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", 1), ("RANK", 0), ("LOCAL_RANK", 0)))
    share = os.environ.get("SLIDE_SHARE_GPU", "0") != "0"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    return rank, world, dev, (torch.device("cpu") if share else dev)


def start_noise(seed, tag, lo, hi, tail, device):
    """x_T rows [lo, hi) of a run: N(0, 1) of shape (hi - lo,) + tail, a function of (seed, tag, GLOBAL shape index) only -- drawn
    in fixed blocks of 1024 shapes, each from its own generator -- so that a shape's start noise does not depend on how the run
    is split into ranks and batches"""
    out = []
    g = torch.Generator(device=device)
    for k in range(lo // 1024, (max(hi, lo + 1) - 1) // 1024 + 1):
        g.manual_seed((int(seed) * 1000003 + int(tag) * 7919 + k) & 0x7FFFFFFFFFFF)
        blk = torch.randn((1024,) + tuple(tail), device=device, generator=g)
        out.append(blk[max(lo - 1024 * k, 0):min(hi - 1024 * k, 1024)])
    return torch.cat(out) if out else torch.empty((0,) + tuple(tail), device=device)


def fps_start_indices(seed, lo, hi, n, device):
    """per-shape draw for the first index of the decode's plain-FPS calls for shapes [lo, hi) of a run: uniform in [0, n) -- the decode
    passes n = 2^30 and every upsampling level reduces the draw modulo ITS candidate count (sample_farthest_points; uniform to 2^-20) --
    a function of (seed, GLOBAL shape index) only (blocks of 1024 shapes, like start_noise) -- the reference draws it from the device generator
    (point_upsample_decoder.py:178-180, random_start_point=True), which makes a decoded cloud depend on the rank / batch split"""
    out = []
    g = torch.Generator(device=device)
    for k in range(lo // 1024, (max(hi, lo + 1) - 1) // 1024 + 1):
        g.manual_seed((int(seed) * 1000003 + 3 * 7919 + k) & 0x7FFFFFFFFFFF)
        blk = torch.randint(0, int(n), (1024,), device=device, generator=g, dtype=torch.int64)
        out.append(blk[max(lo - 1024 * k, 0):min(hi - 1024 * k, 1024)])
    return torch.cat(out) if out else torch.empty((0,), device=device, dtype=torch.int64)


def decode_shard(ae, keypoint, feature, labels, batch_size, device, seed=0, global_offset=0, encode=None):
    """this rank's part of BASELINE configs[4] (the reference decodes per rank too,
    pointnet2/mesh_evaluation.py:113-118): PointAutoencoder.decode of the rank's OWN latents, batch by batch ->
    (n_local, 2048, 6) on `device` (empty shard: (0, 2048, 6)).  keypoint (n_local, 16, 3) / feature (n_local, 16, F) / labels
    (n_local,): tensors or arrays.  The plain-FPS start index of shape g is fps_start_indices(seed, g), so a decoded cloud does not
    depend on how the run is split.  encode(lo, hi, keypoint batch, label batch) -> feature batch, when the features come from
    PointAutoencoder.encode instead (autoencoder_decode_keypoint.py --encode_from); returns (clouds, features) then."""
    kp = torch.as_tensor(keypoint, dtype=torch.float32)
    n = int(kp.shape[0])
    clouds, feats = [], []
    for lo, hi in batches(0, n, batch_size):
        k_ = kp[lo:hi].to(device).contiguous()
        lab = torch.as_tensor(labels[lo:hi]).to(device=device, dtype=torch.int64)
        f_ = encode(lo, hi, k_, lab) if encode is not None else torch.as_tensor(feature[lo:hi], dtype=torch.float32).to(device)
        start = fps_start_indices(seed, global_offset + lo, global_offset + hi, 1 << 30, device)  # reduced modulo N per level
        clouds.append(ae.decode(k_, f_.contiguous(), ts=None, label=lab, fps_start_idx=start))
        feats.append(f_)
    out = torch.cat(clouds) if clouds else torch.empty(0, 2048, 6, device=device)
    if encode is not None:
        return out, (torch.cat(feats) if feats else None)
    return out


def shard_range(num_samples, rank, world_size):
    """contiguous slice [start, end) of rank `rank` (ceil division; trailing ranks may be short or empty)"""
    if world_size <= 1:
        return 0, num_samples
    per = int(np.ceil(num_samples / world_size))
    return min(rank * per, num_samples), min((rank + 1) * per, num_samples)


def batches(start, end, batch_size):
    i = start
    while i < end:
        yield i, min(i + batch_size, end)
        i += batch_size


def all_gather_rows(local, num_samples, world_size, device=None):
    """one collective: every rank contributes its (n_local, ...) rows; returns the (num_samples, ...) concatenation in
    rank order on every rank.  Ranks hold different row counts, so rows are padded to the per-rank capacity."""
    if world_size <= 1:
        return local
    per = int(np.ceil(num_samples / world_size))
    t = torch.as_tensor(local)
    dev = device if device is not None else t.device
    pad = torch.zeros((per,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
    pad[: t.shape[0]] = t.to(dev)
    out = [torch.empty_like(pad) for _ in range(world_size)]
    dist.all_gather(out, pad)
    rows = []
    for r in range(world_size):
        s, e = shard_range(num_samples, r, world_size)
        rows.append(out[r][: e - s])
    return torch.cat(rows, dim=0)


def generate_latents(num_samples, batch_size, labels, run_batch, rank=0, world_size=1, gather_device=None,
                     row_shape=None, row_dtype=torch.float32):
    """Drives `run_batch(labels_np [b], lo, hi) -> (b, ...) tensor/ndarray` over this rank's slice and gathers.
    labels: int array [num_samples].  Returns (latents [num_samples, ...] on every rank, per-sample seconds [n_local]).
    row_shape / row_dtype: shape of one output row, e.g. (16, 51); lets a rank whose slice is EMPTY (more ranks than
    samples) join the all-gather without running a chain just to learn the shape."""
    start, end = shard_range(num_samples, rank, world_size)
    outs, timing = [], []
    for lo, hi in batches(start, end, batch_size):
        t0 = time.time()
        o = torch.as_tensor(run_batch(np.asarray(labels[lo:hi]), lo, hi))
        if o.is_cuda:
            torch.cuda.synchronize(o.device)  # the reference's timer stops without a device sync (mesh_evaluation.py:126)
        timing.extend([(time.time() - t0) / (hi - lo)] * (hi - lo))
        outs.append(o)
    local = torch.cat(outs, dim=0) if outs else None
    if row_shape is None and world_size > 1:
        # one tiny object collective tells a rank with an EMPTY slice (more ranks than samples) the row shape; it does
        # not run a throw-away 1000-step chain to learn it
        shapes = [None] * world_size
        dist.all_gather_object(shapes, None if local is None else (tuple(local.shape[1:]), local.dtype))
        row_shape, row_dtype = next(s_ for s_ in shapes if s_ is not None)
    if local is None:  # a rank with an empty slice still takes part in the all-gather
        local = torch.empty((0,) + tuple(row_shape or ()), dtype=row_dtype, device=gather_device)
    full = all_gather_rows(local, num_samples, world_size, device=gather_device)
    return full, np.asarray(timing)


FIVE_CATEGORIES = (0, 2, 3, 4, 6)  # airplane, cabinet, car, chair, lamp: the categories with released checkpoints (README.md:60)


def category_layout(total, categories=FIVE_CATEGORIES):
    """global sample index -> category for a multi-category generation run: category-major, the first total % n categories
    one sample longer.  Returns [(category, start, end)]."""
    n = len(categories)
    base, extra = divmod(int(total), n)
    out, lo = [], 0
    for i, c in enumerate(categories):
        hi = lo + base + (1 if i < extra else 0)
        out.append((int(c), lo, hi))
        lo = hi
    return out


def category_segments(total, rank=0, world_size=1, categories=FIVE_CATEGORIES):
    """the pieces of rank `rank`'s contiguous shard (shard_range: the reference's per-rank ceil-division slices) that fall into
    each category: [(category, lo, hi)] with global indices.  Every category has its own weight set (per-category
    checkpoints), so a rank runs one chain pair per segment -- at 2048 shapes on 8 ranks a shard spans at most 2."""
    s, e = shard_range(total, rank, world_size)
    return [(c, max(lo, s), min(hi, e)) for c, lo, hi in category_layout(total, categories) if max(lo, s) < min(hi, e)]


def generate_categories(total, run_segment, rank=0, world_size=1, categories=FIVE_CATEGORIES, gather_device=None, row_shape=None,
                        row_dtype=torch.float32):
    """BASELINE configs[3]: `total` shapes over several categories, sharded over the ranks, position then feature DDPM per
    segment with that category's weights, ONE all-gather of the latents at the end.
    run_segment(category, lo, hi) -> (hi - lo, ...) tensor.  Returns (latents [total, ...] on every rank, labels [total])."""
    outs = [torch.as_tensor(run_segment(c, lo, hi)) for c, lo, hi in category_segments(total, rank, world_size, categories)]
    local = torch.cat(outs, dim=0) if outs else None
    if row_shape is None and world_size > 1:
        shapes = [None] * world_size
        dist.all_gather_object(shapes, None if local is None else (tuple(local.shape[1:]), local.dtype))
        row_shape, row_dtype = next(s_ for s_ in shapes if s_ is not None)
    if local is None:
        local = torch.empty((0,) + tuple(row_shape or ()), dtype=row_dtype, device=gather_device)
    full = all_gather_rows(local, total, world_size, device=gather_device)
    labels = np.concatenate([np.full(hi - lo, c, np.int64) for c, lo, hi in category_layout(total, categories)])
    return full, labels


def module_prec_of(prec):
    """arithmetic of the MODULE-level path (autoencoder decode / encode: `SLIDE_MODULE_PREC`) that goes with a CLI's `--prec`:
    "mixed" / "fp16" -> "fp16" (fp16 MFMA operands, fp32 accumulation, GroupNorm statistics and soft-max -- the arithmetic bench.py's
    decode leg times; pinned to the reference's decode by tests/test_hip_modules.py::
    test_autoencoder_decode_fp16_operands_matches_reference), "fp32" / "split" -> "fp32" (exact fp32 MFMA)."""
    return "fp16" if prec in ("mixed", "fp16") else "fp32"


@contextlib.contextmanager
def module_precision(prec):
    """`with module_precision(cli_prec):` -- build AND run module-path models (PointAutoencoder ...) in the arithmetic of
    `module_prec_of(cli_prec)`; restores the environment's SLIDE_MODULE_PREC afterwards."""
    want = module_prec_of(prec)
    prev = os.environ.get("SLIDE_MODULE_PREC")
    os.environ["SLIDE_MODULE_PREC"] = want
    try:
        yield want
    finally:
        if prev is None:
            os.environ.pop("SLIDE_MODULE_PREC", None)
        else:
            os.environ["SLIDE_MODULE_PREC"] = prev


def resolve_prec(prec):
    """(position plan arithmetic, feature plan arithmetic) of a `--prec` value.  "mixed" (the default of the generation CLIs and what
    bench.py times since round 5): the position DDPM in the split arithmetic (fp32-grade: its fp16 plan misses north_star's 1e-3 on
    single forwards), the feature DDPM in fp16 operands / fp32 accumulation (<= 1e-3)."""
    return ("split", "fp16") if prec == "mixed" else (prec, prec)


class CategoryChains:
    """The HIP samplers of one rank for a multi-category run (BASELINE configs[3]): for every segment of the rank's shard a
    position sampler and a feature sampler built from THAT category's weights; the feature chain of a segment is
    conditioned on the positions its position chain generated (README.md:69-73: position DDPM -> feature DDPM).
    weights(category) -> (position state dict, feature state dict)."""

    def __init__(self, total, rank, world_size, pos_cfg, feat_cfg, weights, device, prec="fp16", seed=0, categories=FIVE_CATEGORIES):
        self.total, self.rank, self.world, self.device, self.categories = int(total), rank, world_size, device, categories
        self.segments = category_segments(total, rank, world_size, categories)
        # width of a latent row: known on EVERY rank, also on one whose shard is empty -- all ranks must enter the same
        # collectives in generate() (a rank without chains used to take the object all-gather alone)
        self.cx = 3 + int(feat_cfg["pointnet_config"]["in_fea_dim"])
        self.chains = []
        for k, (c, lo, hi) in enumerate(self.segments):
            sd_p, sd_f = weights(c)
            # the in-kernel noise is keyed on (seed, chain nonce, step, GLOBAL element): every chain has the same seed and a fixed
            # nonce, its sample_offset `lo` tells the shapes apart -- a shape's noise does not depend on the world size
            ps, fs = self._make_chain(pos_cfg, feat_cfg, sd_p, sd_f, hi - lo, prec, seed, seed)
            self.chains.append((c, lo, hi, ps, fs))
        self.seed = int(seed)
        # The GPU runs four hardware queues; the runtime maps streams onto them in creation / first-use order, and two busy chains on
        # one queue serialise while another queue may sit idle (DESIGN.md section 9 item 5a').  With more than two segments (five
        # categories: ten chains) the chains are therefore laid onto FOUR streams explicitly -- feature chain k on stream k mod 4,
        # position chain k on stream (k + 2) mod 4: at most two feature chains per stream, deterministic from run to run.
        if len(self.chains) > 2 and all(hasattr(o_, "stream") for ch_ in self.chains for o_ in ch_[3:5]):
            pool = [self.chains[0][4].stream, self.chains[0][3].stream, self.chains[1][4].stream, self.chains[1][3].stream]
            for k, (_, _, _, ps, fs) in enumerate(self.chains):
                fs.stream = fs.stream2 = pool[k % 4]
                ps.stream = ps.stream2 = pool[(k + 2) % 4]

    def _make_chain(self, pos_cfg, feat_cfg, sd_p, sd_f, n, prec, seed_p, seed_f):
        from .diffusion import FeatureSampler, PositionSampler
        ps = PositionSampler(pos_cfg["pointnet_config"], sd_p, n, self.device, pos_cfg["diffusion_config"], prec=resolve_prec(prec)[0],
                             seed=seed_p, use_graph=False)
        fs = FeatureSampler(feat_cfg["pointnet_config"], sd_f, n, self.device, feat_cfg["standard_diffusion_config"], prec=resolve_prec(prec)[1],
                            seed=seed_f, use_graph=False)
        return ps, fs

    def run(self, steps=None):
        """position chain then feature chain of every segment (all `steps` reverse steps, default the full schedule); the
        position chain of segment k + 1 runs beside the feature chain of segment k (independent streams).
        Returns the rank's latents (n_local, 16, 3 + F) on the device.  (The start noise of shape g is start_noise(seed, ..., g),
        independent of ranks and batches: there is no generator argument any more -- ADVICE r3.)"""
        outs, pending = [], None
        for c, lo, hi, ps, fs in self.chains:
            n = hi - lo
            lab = torch.full((n,), c, dtype=torch.int64, device=self.device)
            ps.begin(lab, start_noise(self.seed, 1, lo, hi, (16, 3), self.device), nonce=1, sample_offset=lo)
            ps.advance(ps.T if steps is None else steps)
            if pending is not None:
                outs.append(pending.state())
            kp = ps.state()
            cx = fs.engine.cx
            fs.begin(lab, kp, start_noise(self.seed, 2, lo, hi, (16, cx), self.device), nonce=2, sample_offset=lo)
            fs.advance(fs.T if steps is None else steps)
            pending = fs
        if pending is not None:
            outs.append(pending.state())
        return torch.cat(outs, dim=0) if outs else torch.empty(0, 16, self.cx, device=self.device)

    def generate(self, steps=None):
        """-> (latents [total, 16, 3 + F] on every rank, labels [total]); one all-gather (RCCL over xGMI; gloo in the tests)"""
        local = self.run(steps)
        base = self.segments[0][1] if self.segments else 0
        return generate_categories(self.total, lambda c, lo, hi: local[lo - base:hi - base], self.rank, self.world, self.categories,
                                   gather_device=self.device, row_shape=(16, self.cx))


def sub_batch_sizes(B, P):
    """P sub-batches of a batch of B (multiples of 8 samples where possible: the row tiles of a launch then spread evenly over
    the 8 XCDs) -- the split bench.py times"""
    P = max(1, min(int(P), int(B)))
    if B % 8 == 0 and B // 8 >= P:
        return [8 * ((B // 8) // P + (1 if i < (B // 8) % P else 0)) for i in range(P)]
    return [B // P + (1 if i < B % P else 0) for i in range(P)]


# The position chain beside the feature sub-batches runs on a stream confined to 11/16 of the compute units (176 of 256;
# slide_stream_create_cu_mask): its wide split-arithmetic launches then leave CUs to the latency-critical feature chains at all
# times (only beside fp16 feature chains with the position plan in the split / fp32 arithmetic: the all-fp16 arrangement loses 3 % with
# it).  Measured in bench.py's arrangement (tools/ab/r05_cumask.sh, two runs each): all CUs 391.5 shapes/s, 240 CUs 392.1, 224 393.7,
# 208 394.1, 192 394.6, 176 394.9, 160 394.3, 128 369.9, 64 275.2 (the chain becomes the long pole).
POS_CU_SHARE = 11.0 / 16.0


class PipelinedGenerator:
    """The arrangement bench.py times, as a generator of latents: per batch of B shapes the feature chain
    runs as P independent sub-batch chains and the position chain as one chain over the batch, all replayed round-robin by one
    library call per advance (EagerChainsSampler); with both DDPMs, the position chain of batch i + 1 runs beside the feature
    chains of batch i.  Sub-batches are objects of the same partition the multi-GPU path uses (independent samples): every
    chain is bit-identical to running it alone (`serial=True` does exactly that, for the tests).
    pos = (pointnet_config, state dict, diffusion_config) or None (key points supplied); feat likewise (standard_diffusion_config)
    or None (positions only)."""

    def __init__(self, B, device, pos=None, feat=None, prec="fp16", seed=0, n_sub=3, serial=False, local_resampling=False,
                 global_offset=0):
        from .diffusion import EagerChainsSampler, FeatureSampler, PositionSampler
        assert pos is not None or feat is not None
        self.B, self.device, self.serial = int(B), device, serial
        # Every chain has the SAME seed and nonce; what tells shapes apart is their GLOBAL index (sample_offset of the chain +
        # row): start noise and in-kernel noise of a shape do not depend on ranks, batches or the sub-batch split
        self.seed, self.g0 = int(seed), int(global_offset)
        self.pos = self.feats = None
        # bench.py's headline arrangement (round 6): with both DDPMs, fp16 feature chains and a wide position plan, the position chain
        # runs over TWO batches at half cadence (one step per two rounds of the feature chains: same shapes per unit time, half the
        # dependent launches beside the feature chains; SLIDE_POS_MULT=1 keeps one batch per chain).  The chains' per-sample arithmetic
        # and noise streams do not depend on the batch they run in.
        wide = (pos is not None and feat is not None and not serial and resolve_prec(prec)[0] != "fp16" and resolve_prec(prec)[1] == "fp16")
        self.pos_mult = 2 if (wide and int(os.environ.get("SLIDE_POS_MULT", "2")) == 2) else 1
        self._pos2 = None
        if self.pos_mult == 2:
            # (built BEFORE the feature samplers, and the one-batch position sampler runs on ITS stream: the runtime maps streams onto the
            #  four hardware queues in creation order, and a position chain that lands on a feature chain's queue serialises with it --
            #  244 instead of 370 shapes/s when a fifth stream shifted the feature chains' queues)
            self._pos2 = PositionSampler(pos[0], pos[1], 2 * self.B, device, pos[2], prec=resolve_prec(prec)[0], seed=seed, use_graph=False,
                                         cu_share=POS_CU_SHARE)
        self._pos_args = (pos, prec, seed)
        if pos is not None and self._pos2 is None:
            self.pos = PositionSampler(pos[0], pos[1], self.B, device, pos[2], prec=resolve_prec(prec)[0], seed=seed, use_graph=False,
                                       cu_share=POS_CU_SHARE if wide else 0.0)
        self.sizes = []
        if feat is not None:
            self.sizes = sub_batch_sizes(self.B, n_sub)
            self.feats = [FeatureSampler(feat[0], feat[1], b, device, feat[2], prec=resolve_prec(prec)[1], seed=seed, use_graph=False,
                                         local_resampling=local_resampling) for i, b in enumerate(self.sizes)]
            self.cx = self.feats[0].engine.cx
        self._Eager = EagerChainsSampler
        self.T = (self.pos or self._pos2 or self.feats[0]).T
        # one slide_run_chains call replays every chain of a group for the SAME number of steps: configurations whose
        # position and feature schedules differ in length advance chain by chain instead
        self._same_T = all(s_.T == self.T for s_ in ([self.pos] if self.pos is not None else []) + (self.feats or []))
        self._have_pos = pos is not None

    def _advance(self, samplers):
        if not samplers:
            return
        if self.serial or not self._same_T:
            for s_ in samplers:
                s_.advance(s_.T)
        else:
            order = samplers[:1] + [s_ for s_ in samplers[1:]]
            self._Eager(order).advance(self.T)

    def _begin_feats(self, lab, kp, x_T, extra, g_lo):
        lo = 0
        for f_, b in zip(self.feats, self.sizes):
            kw = {k_: v_[lo:lo + b] for k_, v_ in extra.items()}
            f_.begin(lab[lo:lo + b], kp[lo:lo + b], x_T[lo:lo + b], nonce=2, sample_offset=g_lo + lo, **kw)
            lo += b

    def run(self, n, labels, keypoints=None, extra=None):
        """n shapes (any count: the last batch is padded): global indices global_offset .. global_offset + n.  labels [n];
        keypoints [n, 16, 3] when there is no position DDPM; extra: per-shape arrays handed to FeatureSampler.begin (local
        re-sampling).  Returns [n, 16, 3] / [n, 16, cx] on the device."""
        B, dev = self.B, self.device
        if self.pos_mult == 2 and self._same_T and self.T % 2 == 0 and n > B:
            return self._run_half_cadence(n, labels, extra)
        if self.pos is None and self._have_pos:  # (a run of at most one batch on a generator built for pairs of batches)
            from .diffusion import PositionSampler
            pos, prec, seed = self._pos_args
            self.pos = PositionSampler(pos[0], pos[1], B, dev, pos[2], prec=resolve_prec(prec)[0], seed=seed, use_graph=False,
                                       stream=self._pos2.stream)
        x_T_pos = lambda lo, hi: start_noise(self.seed, 1, self.g0 + lo, self.g0 + lo + B, (16, 3), dev)
        x_T_feat = lambda lo, hi: start_noise(self.seed, 2, self.g0 + lo, self.g0 + lo + B, (16, self.cx), dev)
        labels = torch.as_tensor(np.asarray(labels), dtype=torch.int64, device=dev)
        pad = lambda t, m: torch.cat([t, torch.zeros((B - m,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)]) if m < B else t
        ranges = list(batches(0, n, B))
        outs, pending = [], None  # pending: (lo, hi) of the batch whose feature chains are in flight
        kp_next = None
        for i in range(len(ranges) + 1):
            group = []
            if i < len(ranges) and self.pos is not None:  # position chain of batch i
                lo, hi = ranges[i]
                self.pos.begin(pad(labels[lo:hi], hi - lo), x_T_pos(lo, hi), nonce=1, sample_offset=self.g0 + lo)
                group.append(self.pos)
            if pending is not None:
                group = self.feats[:1] + group + self.feats[1:]  # (the order bench.py launches them in)
            self._advance(group)
            for s_ in group:
                s_.stream.synchronize()
            if pending is not None:
                plo, phi = pending
                outs.append(torch.cat([f_.state() for f_ in self.feats])[:phi - plo])
                pending = None
            if i < len(ranges):
                lo, hi = ranges[i]
                kp = self.pos.state() if self.pos is not None else pad(torch.as_tensor(keypoints[lo:hi], dtype=torch.float32, device=dev), hi - lo)
                if self.feats is None:
                    outs.append(kp[:hi - lo].clone())
                else:
                    ex = {k_: pad(torch.as_tensor(v_[lo:hi], dtype=torch.float32, device=dev), hi - lo) for k_, v_ in (extra or {}).items()}
                    self._begin_feats(pad(labels[lo:hi], hi - lo), kp, x_T_feat(lo, hi), ex, self.g0 + lo)
                    pending = (lo, hi)
        return torch.cat(outs) if outs else torch.empty(0, 16, self.cx if self.feats else 3, device=dev)


    def _run_half_cadence(self, n, labels, extra):
        """run() for both DDPMs with the position chain over PAIRS of batches at half cadence (bench.py's headline arrangement).
        Time is counted in groups of T rounds of feature steps: the position chain of pair k (batches 2k, 2k + 1) makes T / 2 steps in
        each of the groups 2k and 2k + 1; the feature chains of batch j run in group j + 2, beside the position chain of the next pair."""
        B, dev, T = self.B, self.device, self.T
        pos2 = self._pos2
        labels = torch.as_tensor(np.asarray(labels), dtype=torch.int64, device=dev)
        padn = lambda t, m, cap: torch.cat([t, torch.zeros((cap - m,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)]) if m < cap else t
        ranges = list(batches(0, n, B))
        m = len(ranges)
        npairs = (m + 1) // 2
        outs = []
        kp_pair = None     # key points of the pair whose feature chains are being fed
        pending = None     # (lo, hi) of the batch whose feature chains are in flight
        for g in range(2 * npairs + 2):
            group, every = [], []
            k = g // 2
            if g % 2 == 0 and k < npairs:  # begin the position chain of pair k: shapes [lo, lo + 2 B) of the run
                lo = ranges[2 * k][0]
                hi = min(lo + 2 * B, n)
                pos2.begin(padn(labels[lo:hi], hi - lo, 2 * B), start_noise(self.seed, 1, self.g0 + lo, self.g0 + lo + 2 * B, (16, 3), dev),
                           nonce=1, sample_offset=self.g0 + lo)
            if k < npairs:
                group.append(pos2); every.append(2)
            j = g - 2              # the batch whose feature chains run in this group
            if 0 <= j < m:
                lo, hi = ranges[j]
                kp = kp_pair[(j % 2) * B:(j % 2 + 1) * B]
                ex = {k_: padn(torch.as_tensor(v_[lo:hi], dtype=torch.float32, device=dev), hi - lo, B) for k_, v_ in (extra or {}).items()}
                self._begin_feats(padn(labels[lo:hi], hi - lo, B), kp, start_noise(self.seed, 2, self.g0 + lo, self.g0 + lo + B, (16, self.cx), dev),
                                  ex, self.g0 + lo)
                pending = (lo, hi)
                group = self.feats[:1] + group + self.feats[1:]   # (the order bench.py launches them in)
                every = [1] + every + [1] * (len(self.feats) - 1)
            if group:
                self._Eager(group, every=every).advance(T)
                for s_ in group:
                    s_.stream.synchronize()
            if pending is not None:
                plo, phi = pending
                outs.append(torch.cat([f_.state() for f_ in self.feats])[:phi - plo])
                pending = None
            if g % 2 == 1 and k < npairs:  # the pair's position chain has made its T steps
                kp_pair = pos2.state()
        return torch.cat(outs) if outs else torch.empty(0, 16, self.cx, device=dev)


def save_generated(save_dir, points, labels, timing, num_points, keypoint=None, keypoint_feature=None, ckpt_info=""):
    """npz schema of mesh_evaluation.py:135-150: points,label,category,category_name,timing[,keypoint,keypoint_feature]"""
    os.makedirs(save_dir, exist_ok=True)
    f = os.path.join(save_dir, "shapenet_psr_generated_data_%d_pts%s.npz" % (num_points, ckpt_info))
    labels = np.asarray(labels)
    d = {"points": np.asarray(points), "label": labels, "category": [CATEGORY_IDS[int(i)] for i in labels],
         "category_name": [CATEGORY_NAMES[int(i)] for i in labels], "timing": np.asarray(timing)}
    if keypoint is not None:
        d["keypoint"] = np.asarray(keypoint)
    if keypoint_feature is not None:
        d["keypoint_feature"] = np.asarray(keypoint_feature)
    np.savez(f, **d)
    return f
