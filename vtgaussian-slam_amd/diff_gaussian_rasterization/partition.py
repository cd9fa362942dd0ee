"""Tile-row image partition for multi-GPU rendering: contiguous bands of 16-pixel tile rows,
Gaussians replicated, one band per rank; the collectives of the tracking loop (7-float pose gradient) and of the mapping
loop (per-Gaussian gradients, SSIM halo rows, global mask count and median)."""
from typing import List, Tuple


def tile_rows_total(image_height: int) -> int:
    return (int(image_height) + 15) // 16


def band_for_rank(image_height: int, world_size: int, rank: int) -> Tuple[int, int]:
    """(begin, end) tile rows of `rank`; the first `rows % world_size` ranks get one extra row.
    Raises if there are more ranks than tile rows (an empty band is not a valid render)."""
    rows = tile_rows_total(image_height)
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    if world_size > rows:
        raise ValueError(f"{world_size} ranks for {rows} tile rows: reduce the number of GPUs")
    per, rem = divmod(rows, world_size)
    b = rank * per + min(rank, rem)
    return b, b + per + (1 if rank < rem else 0)


def all_bands(image_height: int, world_size: int) -> List[Tuple[int, int]]:
    return [band_for_rank(image_height, world_size, r) for r in range(world_size)]


def pixel_rows(band: Tuple[int, int], image_height: int) -> Tuple[int, int]:
    return band[0] * 16, min(band[1] * 16, int(image_height))


class OwnedSet:
    """The Gaussians of the map that can meet one rank's band of tile rows (include/vtgs.h, "Owned sets").

    Every rank of the partition holds the whole map, but 7 of 8 Gaussians cannot meet its rows; without a list each rank still
    runs the per-Gaussian kernels (pose transform, projection + binning, gradient gather) over all of them.  The set is built
    with the band test of the projection kernel widened by `margin_px` pixels and `growth` x the scales, so that it stays valid
    while the pose and the scales move a little (a tracking or mapping phase); `fused.render_frame(..., owned=set)` then hands
    the rasterizer compact arrays of the listed Gaussians.

    Validity is CHECKED, not assumed: before every render `check` runs the exact band test (1 px of slack for float32
    rounding) over the whole map -- 16 bytes read per Gaussian -- and counts on the device the Gaussians outside the list that
    could now meet the band.  While `escaped()` (one device read; call it at the end of a phase) is 0, every render of the
    phase equals the band render of the whole map: image, radii and per-Gaussian gradients bit for bit (the list is ascending,
    so equal depths sort as before), the pose gradient to float32 rounding (its partial sums group the Gaussians by rows of
    256 of the list instead of the map).  A non-zero count means the margin was too small for how far the phase moved:
    rebuild and redo the phase.
    """

    def __init__(self, params, time_idx: int, raster_settings, first_frame_w2c, band: Tuple[int, int], margin_px: float = 32.0,
                 growth: float = 1.25, radius_rule=None, with_centre_rows: bool = False):
        """The list for the fused caller chain: world-frame `params` under the pose of frame `time_idx`
        (`fused.render_frame(..., owned=set)`).  `first_frame_w2c` is accepted for symmetry with render_frame; the band test
        does not need it."""
        import torch
        _need_device(params["means3D"], "OwnedSet")
        if params["log_scales"].shape[1] != 1:
            raise RuntimeError("owned sets are built for isotropic maps (log_scales [N,1], every reference config)")
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        self._build(raster_settings, band, margin_px, growth, radius_rule, f32(params["means3D"]), f32(params["log_scales"]), True,
                    f32(params["cam_unnorm_rots"][0, :, time_idx]), f32(params["cam_trans"][0, :, time_idx]), with_centre_rows)

    @classmethod
    def for_operator(cls, means3D, scales, raster_settings, band: Tuple[int, int], margin_px: float = 32.0, growth: float = 1.25,
                     radius_rule=None):
        """The list for the plain operator: camera-frame `means3D` [N,3] and `scales` [N,3] as `GaussianRasterizer.forward`
        takes them (`GaussianRasterizer(raster_settings, tile_rows=band, owned=set)`)."""
        import torch
        _need_device(means3D, "OwnedSet")
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        self = cls.__new__(cls)
        self._build(raster_settings, band, margin_px, growth, radius_rule, f32(means3D), f32(scales), False, None, None, False)
        return self

    def _build(self, raster_settings, band, margin_px, growth, radius_rule, means3D, scales, scales_are_log, cam_q, cam_t,
               with_centre_rows=False):
        import os

        import torch

        from . import _RADIUS_RULES, _Camera
        self.band = (int(band[0]), int(band[1]))
        self.margin_px, self.growth = float(margin_px), float(growth)
        if self.margin_px < 1.0 or self.growth < 1.0:          # the check tests with 1 px of slack: a tighter list fails it at once
            raise ValueError("OwnedSet needs margin_px >= 1 and growth >= 1")
        dev = means3D.device
        rule = _RADIUS_RULES[radius_rule or os.environ.get("VTGS_RADIUS_RULE", "3sigma")]
        cam = _Camera(raster_settings, dev, rule, self.band)
        self.n_map, self.scales_are_log = int(means3D.shape[0]), bool(scales_are_log)
        self.mask = torch.empty(self.n_map, dtype=torch.uint8, device=dev)
        self.escapes = torch.zeros(1, dtype=torch.int32, device=dev)
        # centre_rows: the 16-pixel tile row of every Gaussian's projected centre (-1 behind the near plane) -- its owner band
        self.centre_rows = torch.empty(self.n_map, dtype=torch.int32, device=dev) if with_centre_rows else None
        self._test(cam, means3D, scales, cam_q, cam_t, self.margin_px, self.growth, None, self.mask, None, None, self.centre_rows)
        self.idx64 = torch.nonzero(self.mask).reshape(-1)              # ascending; one device -> host read (the count)
        self.idx = self.idx64.to(torch.int32)

    def __len__(self) -> int:
        return int(self.idx.numel())

    def _test(self, cam, means3D, scales, cam_q, cam_t, margin_px, growth, owned, mask_out, escapes, stream, centre_rows=None):
        import ctypes

        from . import _check, _lib, _stream_ptr
        ptr = lambda x: None if x is None else x.data_ptr()
        _check(_lib.vtgs_band_owner_mask(ctypes.byref(cam.c), means3D.shape[0], means3D.data_ptr(), scales.data_ptr(),
                                         1 if self.scales_are_log else 0, ptr(cam_q), ptr(cam_t), margin_px, growth, ptr(owned),
                                         ptr(mask_out), ptr(escapes), ptr(centre_rows),
                                         _stream_ptr(means3D.device) if stream is None else stream),
               "vtgs_band_owner_mask")

    def admit(self, n_map: int, band) -> None:
        """Is this the band and the map the list was built for?  (host-side, before a render)"""
        if band is None or (int(band[0]), int(band[1])) != self.band:
            raise ValueError(f"owned set built for tile rows {self.band}, render asks for {band}")
        if int(n_map) != self.n_map:
            raise ValueError(f"owned set built for a map of {self.n_map} Gaussians, the map has {int(n_map)} "
                             "(rebuild after densification / pruning)")

    def check(self, cam, means3D, scales, cam_q=None, cam_t=None, stream=None) -> None:
        """Enqueue the exact band test of the whole map under the pose / scales of the render about to run (float32 contiguous
        device tensors, as the autograd nodes hold them; no host wait)."""
        self._test(cam, means3D, scales, cam_q, cam_t, 1.0, 1.0, self.mask, None, self.escapes, stream)

    def escaped(self) -> int:
        """Gaussians outside the list that could have met the band in some render since the set was built (device read)."""
        return int(self.escapes.item())


def phase_escapes(owned_sets, group=None) -> int:
    """End of a phase on N ranks: the escape counters of this rank's lists, summed over ALL ranks (one small all-reduce), so
    that every rank takes the same decision -- 0: the phase stands; otherwise every rank rebuilds its lists (wider margin) and
    redoes the phase.  A rank deciding alone would leave the others waiting in the next collective."""
    import torch
    import torch.distributed as dist
    sets = list(owned_sets)
    dev = sets[0].escapes.device if sets else "cpu"
    total = torch.zeros(1, dtype=torch.float32, device=dev)
    for o in sets:
        total += o.escapes.to(torch.float32)
    if dist.is_available() and dist.is_initialized():
        all_reduce_sum(total, group)
    return int(total.item())


class PhaseSnapshot:
    """What "redo the phase" needs.  With deferred run-ahead overflows the iteration that overflowed still ran to its
    end on every rank: the overflowing rank rendered the BACKGROUND for its band, its gradients went into the collectives, and
    `optimizer.step()` applied them -- and so did every later iteration of the phase, until `phase_overflows` raises.  Redoing the
    phase is only correct from the state BEFORE it: take a snapshot of every tensor the phase's optimizer moves (and nothing
    else: the tracking phase moves the two camera tensors, 7 T floats; the mapping phase the trainable Gaussian groups), run the
    phase, and on the error `restore()` and build a fresh optimizer (its moments belong to the discarded steps).

        snap = PhaseSnapshot(params, ("cam_unnorm_rots", "cam_trans"))
        try:
            run_the_phase(); partition.phase_overflows()
        except partition.RunAheadOverflow:
            snap.restore(); optimizer = make_optimizer(); run_the_phase(); partition.phase_overflows()

    bench_slam.py's N-rank loop does exactly this (one redo: the capacities were raised where it happened)."""

    def __init__(self, params, keys):
        self.params, self.saved = params, {k: params[k].detach().clone() for k in keys}

    def restore(self) -> None:
        import torch
        with torch.no_grad():
            for k, v in self.saved.items():
                self.params[k].copy_(v)
                self.params[k].grad = None


class RunAheadOverflow(RuntimeError):
    """Raised by `phase_overflows` on EVERY rank together."""


def phase_overflows(group=None, device=None) -> int:
    """End of a phase on N ranks, with `diff_gaussian_rasterization.defer_run_ahead_overflow(True)`: the run-ahead overflows the
    ranks recorded during the phase, summed over all ranks (one small all-reduce), and the SAME error (`RunAheadOverflow`) on
    every rank when the sum is not zero -- a rank raising alone, inside its `backward()`, would leave the others in the next
    collective.  The capacities have already been raised where it happened; the parameters and the optimizer
    state are those of a phase that consumed invalid gradients: restore them (`PhaseSnapshot`) and redo the phase.
    `device`: where the one-float count lives for the all-reduce; default: the current HIP device when the group's backend
    reduces device tensors (RCCL), the host otherwise (gloo)."""
    import torch
    import torch.distributed as dist
    import diff_gaussian_rasterization as dgr
    dgr.settle_pending()                              # (deferred: records, does not raise)
    if device is None:
        device = "cpu"
        if dist.is_available() and dist.is_initialized() and not _host_staged(group) and torch.cuda.is_available():
            device = torch.device("cuda", torch.cuda.current_device())
    total = torch.tensor([float(dgr.deferred_overflows())], dtype=torch.float32, device=device)
    if dist.is_available() and dist.is_initialized():
        all_reduce_sum(total, group)
    n = int(total.item())
    if n:
        raise RunAheadOverflow(f"{n} run-ahead forward(s) overflowed their workspace on some rank during this phase: the images and "
                               "gradients of those iterations were invalid there and the optimizer has consumed them.  Capacities "
                               "have been raised; restore the phase's parameters (partition.PhaseSnapshot) and redo the phase "
                               "(every rank raises this together)")
    return 0


class OwnerExchange:
    """Mapping on N ranks without the 20 N-byte all-reduce ("reduce-scatter by Gaussian owner band ... sparse halo
    exchange with the neighbouring bands only, then each GPU runs Adam on its owned slice").

    Every Gaussian has ONE owner for the phase: the rank whose band holds the tile row of its projected centre under the phase's
    reference pose (`OwnedSet.centre_rows`: computed from the same bytes on every rank, so every rank names the same owner).
    `lists` is this rank's UNION list for the phase -- an `OwnedSet` built with `with_centre_rows=True` and a margin that covers
    every view the phase renders (the per-view lists the renders use must be subsets of it: `covers()`).  Built collectively
    (one all-gather of the ranks' N-byte masks).  Per iteration, after `backward()`:

        ex.reduce_grads(params)        # halo rows -> their owners (point-to-point, only between ranks that share Gaussians)
        optimizer.step(rows=ex.update_rows)   # Adam on the rows this rank owns (+ the rows nobody lists, replicated)
        ex.publish(params)             # updated rows -> the ranks that list them

    and `ex.gather_all(params)` at the end of the phase (or before lists are rebuilt) hands every rank every owner's rows.
    Between those points a rank holds current values for the rows of ITS list only -- all it renders from.
    The gradient of a row is its owner's own contribution plus the listers' in ascending rank order: the same on every run.
    """

    def __init__(self, lists, image_height: int, rank: int, world: int, keys=("rgb_colors", "logit_opacities", "log_scales"),
                 group=None, frozen=("means3D", "unnorm_rotations")):
        import torch
        import torch.distributed as dist
        if lists.centre_rows is None:
            raise ValueError("OwnerExchange needs an OwnedSet built with with_centre_rows=True")
        self.rank, self.world, self.group, self.keys = int(rank), int(world), group, tuple(keys)
        self._frozen_ok = frozenset(frozen)      # per-Gaussian parameters the caller steps with learning rate 0 (not exchanged)
        dev = lists.mask.device
        self.n_map = lists.n_map
        ends = torch.tensor([e for _b, e in all_bands(image_height, world)], device=dev, dtype=torch.int32)
        owner = torch.bucketize(lists.centre_rows.clamp(min=0), ends, right=True).clamp(max=world - 1)     # [N], same on every rank
        staged = lists.mask.is_cuda and _host_staged(group)
        mine = lists.mask.cpu() if staged else lists.mask
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine, group=group)
        masks = torch.stack(gathered).to(dev).bool()                                                      # [world, N]
        listed = masks.any(0)
        own = (owner == rank) & listed
        if bool((own & ~masks[rank]).any()):
            raise RuntimeError("a Gaussian owned by this rank's band is missing from its list (margin below the band test's slack?)")
        self.own_rows = torch.nonzero(own).reshape(-1)
        self.update_rows = torch.nonzero(own | ~listed).reshape(-1).to(torch.int32)      # + the rows nobody lists: replicated Adam
        self.listed = listed
        self.union_mask = masks[rank]
        # grads travel lister -> owner, parameters owner -> lister: the same two index sets per pair of ranks
        self.to_owner = {q: torch.nonzero(masks[rank] & (owner == q)).reshape(-1) for q in range(world) if q != rank}
        self.from_lister = {q: torch.nonzero(masks[q] & own).reshape(-1) for q in range(world) if q != rank}
        self.to_owner = {q: v for q, v in self.to_owner.items() if v.numel()}
        self.from_lister = {q: v for q, v in self.from_lister.items() if v.numel()}
        self.halo_rows = int(sum(v.numel() for v in self.to_owner.values()))

    def covers(self, owned) -> bool:
        """Is the per-view list `owned` (what a render of this phase uses) inside this rank's union list?  (device read)"""
        return not bool((owned.mask.bool() & ~self.union_mask).any())

    def _flat(self, params, rows, grad: bool):
        import torch
        return torch.cat([(params[k].grad if grad else params[k].detach()).reshape(self.n_map, -1).index_select(0, rows)
                          for k in self.keys], dim=1)

    def _exchange(self, send, recv_rows, width, dev):
        import torch
        import torch.distributed as dist
        staged = dev.type == "cuda" and _host_staged(self.group)
        recv = {q: torch.empty((n, width), dtype=torch.float32, device="cpu" if staged else dev) for q, n in recv_rows.items()}
        keep = {q: (t.cpu() if staged else t.contiguous()) for q, t in send.items()}
        # (P2POp takes GLOBAL ranks; q is a rank of `group`: ADVICE r4)
        peer = (lambda q: q) if self.group is None else (lambda q: dist.get_global_rank(self.group, q))
        ops = [dist.P2POp(dist.irecv, recv[q], peer(q), self.group) for q in sorted(recv)]
        ops += [dist.P2POp(dist.isend, keep[q], peer(q), self.group) for q in sorted(keep)]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return {q: r.to(dev) for q, r in recv.items()}

    def reduce_grads(self, params) -> int:
        """Send the gradients of the halo rows to their owners and add what the listers of this rank's rows send (in place on
        `.grad`; a parameter without a gradient on this rank counts as zero).  Returns the bytes this rank sent."""
        import torch
        dev = params[self.keys[0]].device
        # A per-Gaussian parameter that is NOT exchanged but has a gradient would be stepped from this rank's band share alone
        # and drift apart across the ranks (harmless only while its learning rate is 0, as for means3D / unnorm_rotations in
        # the reference's mapping configs): refuse it here, once per call, instead of diverging silently.
        for k, v in params.items():
            if k not in self.keys and torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == self.n_map and v.grad is not None \
                    and k not in self._frozen_ok:
                raise RuntimeError(f"OwnerExchange: '{k}' has a per-Gaussian gradient but is not among the exchanged keys {self.keys}; "
                                   f"add it to keys, or declare it frozen (learning rate 0) with OwnerExchange(..., frozen=('{k}',))")
        for k in self.keys:
            if params[k].grad is None:
                params[k].grad = torch.zeros_like(params[k])
        width = sum(int(params[k].numel() // self.n_map) for k in self.keys)
        send = {q: self._flat(params, rows, True) for q, rows in self.to_owner.items()}
        got = self._exchange(send, {q: int(r.numel()) for q, r in self.from_lister.items()}, width, dev)
        for q in sorted(got):                                           # ascending rank: a fixed summation order
            o = 0
            for k in self.keys:
                w = int(params[k].numel() // self.n_map)
                params[k].grad.reshape(self.n_map, -1).index_add_(0, self.from_lister[q], got[q][:, o:o + w])
                o += w
        return sum(int(t.numel()) * 4 for t in send.values())

    def publish(self, params) -> int:
        """After the owner's Adam step: the updated rows go to the ranks that list them (in place on the parameters)."""
        import torch
        dev = params[self.keys[0]].device
        width = sum(int(params[k].numel() // self.n_map) for k in self.keys)
        send = {q: self._flat(params, rows, False) for q, rows in self.from_lister.items()}
        got = self._exchange(send, {q: int(r.numel()) for q, r in self.to_owner.items()}, width, dev)
        with torch.no_grad():
            for q, buf in got.items():
                o = 0
                for k in self.keys:
                    w = int(params[k].numel() // self.n_map)
                    params[k].reshape(self.n_map, -1).index_copy_(0, self.to_owner[q], buf[:, o:o + w])
                    o += w
        return sum(int(t.numel()) * 4 for t in send.values())

    def gather_all(self, params) -> None:
        """End of the phase: every rank receives every owner's rows (one all-reduce of the owners' disjoint contributions);
        the rows nobody lists were updated on every rank alike."""
        import torch
        with torch.no_grad():
            width = sum(int(params[k].numel() // self.n_map) for k in self.keys)
            flat = torch.zeros((self.n_map, width), dtype=torch.float32, device=params[self.keys[0]].device)
            flat.index_copy_(0, self.own_rows, self._flat(params, self.own_rows, False))
            all_reduce_sum(flat, self.group)
            o = 0
            for k in self.keys:
                w = int(params[k].numel() // self.n_map)
                p2 = params[k].reshape(self.n_map, -1)
                p2.copy_(torch.where(self.listed[:, None], flat[:, o:o + w], p2))
                o += w


def _host_staged(group=None) -> bool:
    """gloo moves host memory: device tensors go through a CPU copy (the rehearsal of the N-rank path on one GPU box and the
    CPU tests); RCCL (backend "nccl") takes device tensors as they are."""
    import torch.distributed as dist
    return dist.get_backend(group) == "gloo"


def all_reduce_sum(t, group=None):
    """In-place sum over the ranks of a (small) tensor on any device, with either backend."""
    import torch.distributed as dist
    if t.is_cuda and _host_staged(group):
        c = t.cpu()
        dist.all_reduce(c, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, group=group)
    return t



def pose7_reduce(points, g_points):
    """{sum g, sum p x g, sum g_z} of camera-frame points and their gradients as ONE 7-float device tensor (two launches):
    what every rank of the tile-row partition contributes to the pose-gradient all-reduce."""
    import ctypes

    import torch

    from . import _check, _lib, _stream_ptr
    if not points.is_cuda:
        raise RuntimeError("pose7_reduce needs tensors on a HIP device (torch 'cuda'); no CPU path exists")
    p = points.detach().to(torch.float32).contiguous()
    g = g_points.detach().to(torch.float32).contiguous()
    n = p.shape[0]
    rows = int(_lib.vtgs_pose_partial_rows(n))
    partials = torch.empty((max(rows, 1), 7), dtype=torch.float32, device=p.device)
    out = torch.empty(7, dtype=torch.float32, device=p.device)
    _check(_lib.vtgs_pose7_reduce(n, p.data_ptr(), g.data_ptr(), partials.data_ptr(), out.data_ptr(), _stream_ptr(p.device)),
           "vtgs_pose7_reduce")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Mapping mode of the tile-row partition ("non-additive pieces").  Tracking needs one 7-float all-reduce per
# iteration (pose7_reduce above); mapping optimises the Gaussians themselves, so
#   * the trainable per-Gaussian gradients (rgb 3 + opacity 1 + log-scale 1 = 20 B per Gaussian; means and rotations have
#     learning rate 0, configs/replica/room0.py:100-102) are summed over the ranks in ONE flat all-reduce per iteration;
#   * the SSIM term has an 11x11 window (utils/slam_external.py:78-87): a band needs 5 rendered pixel rows of each
#     neighbour (HaloExchange: a differentiable send/recv pair, the halo's gradient flows back to the rank that rendered it);
#   * the depth term is a masked MEAN: every rank divides its partial sum by the GLOBAL mask count (src/vtgaussian_slam.py:
#     592-597), and the 50 x median outlier mask (:525-528, :754) needs the median over the whole frame (global_median).
# All of it is torch + torch.distributed on whatever device the tensors live on: RCCL (backend "nccl") on the GPUs, gloo in
# the CPU tests (tests/test_band_partition_gloo.py, world_size 2).
# ---------------------------------------------------------------------------------------------------------------------
SSIM_HALO = 5                      # rows a band needs from each neighbour: (11 - 1) / 2


def _need_device(t, what: str) -> None:
    """The band losses are HIP kernels (csrc/vtgs_loss.hip); this package has no CPU path.  (Round 3 carried pure-torch CPU
    branches here for the gloo tests: they live in tests/band_cpu_ref.py now, on top of this module's collectives.)"""
    if not t.is_cuda:
        raise RuntimeError(f"{what} needs tensors on a HIP device (torch 'cuda'); no CPU path exists")


def allreduce_radii(radii, group=None):
    """A rank of the partition skips the Gaussians that cannot meet its rows (radius 0 there): the radii of the frame --
    variables['max_2D_radius'] / ['seen'] of the reference, src/vtgaussian_slam.py:681-689 -- are the element-wise maximum
    over the ranks.  In place; returns `radii`."""
    import torch.distributed as dist
    if radii.is_cuda and _host_staged(group):
        c = radii.cpu()
        dist.all_reduce(c, op=dist.ReduceOp.MAX, group=group)
        radii.copy_(c)
    else:
        dist.all_reduce(radii, op=dist.ReduceOp.MAX, group=group)
    return radii


def allreduce_param_grads(params, keys=("rgb_colors", "logit_opacities", "log_scales"), group=None):
    """Sum `.grad` of the listed parameters over the ranks with one flat all-reduce (20 B per Gaussian for the default
    keys).  Parameters without a gradient on this rank (no Gaussian of theirs met the band) count as zero."""
    import torch
    import torch.distributed as dist
    ref = params[keys[0]]
    parts = [(params[k].grad if params[k].grad is not None else torch.zeros_like(params[k])).reshape(-1) for k in keys]
    flat = torch.cat(parts).to(ref.device)
    all_reduce_sum(flat, group)
    o = 0
    for k, p in zip(keys, parts):
        n = p.numel()
        g = flat[o:o + n].reshape(params[k].shape)
        if params[k].grad is None:
            params[k].grad = g.clone()
        else:
            params[k].grad.copy_(g)
        o += n
    return flat.numel() * flat.element_size()


def _halo_autograd():
    import torch
    import torch.distributed as dist

    def _exchange(ops, recv_up, recv_down, group):
        """Runs the batched sends / receives; with gloo and device tensors the payload goes through host copies."""
        if not ops:
            return recv_up, recv_down
        staged = any(o.tensor.is_cuda for o in ops) and _host_staged(group)
        if staged:
            host = {id(o.tensor): o.tensor.cpu() for o in ops}
            ops = [dist.P2POp(o.op, host[id(o.tensor)], o.peer, group) for o in ops]
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        if staged:
            recv_up = None if recv_up is None else host[id(recv_up)].to(recv_up.device)
            recv_down = None if recv_down is None else host[id(recv_down)].to(recv_down.device)
        return recv_up, recv_down

    class HaloExchange(torch.autograd.Function):
        """img [C,H,W] holds this rank's pixel rows [y0, y1) (anything elsewhere is ignored).  Returns a copy whose rows
        [y0 - halo, y0) and [y1, y1 + halo) hold what the neighbouring ranks rendered there.  Backward: the gradient
        that lands on a halo row is sent to the rank that rendered it and added to that rank's own-row gradient."""

        @staticmethod
        def forward(ctx, img, y0: int, y1: int, halo: int, rank: int, world: int, group):
            H = img.shape[-2]
            out = img.clone()
            up, down = rank - 1, rank + 1                       # bands are ordered top to bottom by rank
            lo0, hi1 = max(y0 - halo, 0), min(y1 + halo, H)
            ops, recv_up, recv_down = [], None, None
            if up >= 0:
                send = img[:, y0:min(y0 + halo, y1)].contiguous()
                recv_up = torch.empty_like(img[:, lo0:y0])
                ops += [dist.P2POp(dist.isend, send, up, group), dist.P2POp(dist.irecv, recv_up, up, group)]
            if down < world:
                send = img[:, max(y1 - halo, y0):y1].contiguous()
                recv_down = torch.empty_like(img[:, y1:hi1])
                ops += [dist.P2POp(dist.isend, send, down, group), dist.P2POp(dist.irecv, recv_down, down, group)]
            recv_up, recv_down = _exchange(ops, recv_up, recv_down, group)
            if recv_up is not None:
                out[:, lo0:y0] = recv_up
            if recv_down is not None:
                out[:, y1:hi1] = recv_down
            ctx.cfg = (y0, y1, halo, rank, world, group, H)
            return out

        @staticmethod
        def backward(ctx, g):
            y0, y1, halo, rank, world, group, H = ctx.cfg
            up, down = rank - 1, rank + 1
            lo0, hi1 = max(y0 - halo, 0), min(y1 + halo, H)
            gi = torch.zeros_like(g)
            gi[:, y0:y1] = g[:, y0:y1]
            ops, recv_up, recv_down = [], None, None
            if up >= 0:                                         # my top halo was rendered by `up`; `up`'s bottom halo by me
                send = g[:, lo0:y0].contiguous()
                recv_up = torch.empty_like(g[:, y0:min(y0 + halo, y1)])
                ops += [dist.P2POp(dist.isend, send, up, group), dist.P2POp(dist.irecv, recv_up, up, group)]
            if down < world:
                send = g[:, y1:hi1].contiguous()
                recv_down = torch.empty_like(g[:, max(y1 - halo, y0):y1])
                ops += [dist.P2POp(dist.isend, send, down, group), dist.P2POp(dist.irecv, recv_down, down, group)]
            recv_up, recv_down = _exchange(ops, recv_up, recv_down, group)
            if recv_up is not None:
                gi[:, y0:min(y0 + halo, y1)] += recv_up
            if recv_down is not None:
                gi[:, max(y1 - halo, y0):y1] += recv_down
            return gi, None, None, None, None, None, None
    return HaloExchange


def halo_exchange(img, band: Tuple[int, int], image_height: int, rank: int, world: int, halo: int = SSIM_HALO, group=None):
    """Differentiable exchange of `halo` pixel rows with the neighbouring bands (see HaloExchange)."""
    y0, y1 = pixel_rows(band, image_height)
    if y1 - y0 < halo and world > 1:
        raise ValueError(f"a band of {y1 - y0} pixel rows cannot serve a {halo}-row halo; use fewer ranks")
    return _halo_autograd().apply(img, y0, y1, halo, rank, world, group)


def global_median(local_values, group=None):
    """torch.median (the lower median, as the reference's depth_error.median()) of the concatenation of every rank's values."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = local_values.device
    v = local_values.detach().reshape(-1)
    if v.is_cuda and _host_staged(group):
        return global_median(v.cpu(), group).to(dev)
    counts = [torch.zeros(1, dtype=torch.long, device=v.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([v.numel()], dtype=torch.long, device=v.device), group=group)
    m = int(max(int(c) for c in counts))
    pad = torch.full((m,), float("nan"), dtype=v.dtype, device=v.device)
    pad[:v.numel()] = v
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:int(c)] for b, c in zip(bufs, counts)]).median()


def band_mapping_loss(im, depth_sil, gt_im, gt_depth, band: Tuple[int, int], rank: int, world: int, w_im: float = 1.0,
                      w_depth: float = 1.0, ignore_outlier_depth_loss: bool = False, group=None):
    """This rank's share of the mapping loss of get_loss (src/vtgaussian_slam.py:519-611): the shares of all ranks sum to the
    full-frame loss and the gradients of the shares, summed over the ranks (allreduce_param_grads), are the full-frame
    gradients.  im / depth_sil: [3,H,W] renders whose rows of `band` are this rank's (GaussianRasterizer(tile_rows=band));
    gt_im / gt_depth: full frames (every rank holds the ground truth)."""
    import torch
    import torch.distributed as dist
    import torch.nn.functional as F
    H, W = im.shape[-2], im.shape[-1]
    y0, y1 = pixel_rows(band, H)
    rows = slice(y0, y1)
    _need_device(im, "band_mapping_loss")
    # the band forms of the loss kernels (vtgs_slam_loss_band_*): SSIM over the band's rows with the neighbours' rows as
    # context, masked sums, ONE 8-float all-reduce (global mask count), gradient images in 2 launches
    from . import losses as _l
    full = halo_exchange(im, band, H, rank, world, SSIM_HALO, group) if world > 1 else im
    extra = None
    if ignore_outlier_depth_loss:
        gd, d = gt_depth[:, rows], depth_sil[0:1, rows].detach()
        err = torch.abs(gd - d) * (gd > 0)
        med = global_median(err, group) if world > 1 else err.median()
        extra = torch.zeros((H, W), dtype=torch.float32, device=im.device)
        extra[rows] = ((err < 50 * med) & (gd > 0))[0].to(torch.float32)
    red = (lambda t: all_reduce_sum(t, group)) if world > 1 else None
    return _l.band_loss(full, depth_sil, gt_im, gt_depth, (y0, y1), "mapping", w_im=w_im, w_depth=w_depth,
                        extra_mask=extra, reduce=red, first_band=(rank == 0))


def band_tracking_loss(im, depth_sil, gt_im, gt_depth, band: Tuple[int, int], sil_thres: float, w_im: float = 0.5,
                       w_depth: float = 0.025, extra_mask=None, colour_over_all_pixels: bool = False):
    """This rank's share of the tracking loss of get_loss (src/vtgaussian_slam.py:519-605): masked SUMS, so the shares of
    all ranks add up to the full-frame loss without any exchange; the pose gradient is all-reduced afterwards (7 floats).
    extra_mask: full-frame [H,W] / [1,H,W] (the TUM / ScanNet masks), only the band's rows are read."""
    import torch
    H = im.shape[-2]
    y0, y1 = pixel_rows(band, H)
    _need_device(im, "band_tracking_loss")
    from . import losses as _l
    return _l.band_loss(im, depth_sil, gt_im, gt_depth, (y0, y1), "tracking", sil_thres=sil_thres, w_im=w_im,
                        w_depth=w_depth, extra_mask=extra_mask, colour_over_all_pixels=colour_over_all_pixels)


def band_silhouette_threshold(im, silhouette, gt_im, gt_depth, band: Tuple[int, int], world: int,
                              candidates=(0.990, 0.993, 0.995, 0.997, 0.999), group=None, sums=None):
    """The threshold pick of tracking iteration 0 (src/vtgaussian_slam.py:472-510) over a partitioned frame: every rank
    sums its band's squared error and pixel count per candidate (one kernel), ONE all-reduce of 2 x K floats, the same
    arg-min on every rank.  `sums` ([K, 2] float64: squared error, pixel count per candidate over the band) lets a caller
    that formed the band's sums elsewhere use the reduction and the pick alone (the CPU tests do, tests/band_cpu_ref.py)."""
    H = im.shape[-2]
    y0, y1 = pixel_rows(band, H)
    if sums is None:
        _need_device(im, "band_silhouette_threshold")
        from . import losses as _l
        sums = _l.silhouette_sweep(im, silhouette, gt_im, gt_depth, candidates, rows=(y0, y1))
    if world > 1:
        all_reduce_sum(sums, group)
    sums = sums.cpu()
    best, best_mse = candidates[0], float("inf")
    for k, c in enumerate(candidates):
        cnt = float(sums[k, 1])
        mse = float(sums[k, 0]) / (3.0 * cnt) if cnt > 0 else float("inf")
        if mse < best_mse:
            best, best_mse = c, mse
    return best
