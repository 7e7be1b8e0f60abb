"""InstanceBank: the learned anchor/feature tables plus the recurrent temporal state of the decoder
(one state row per camera stream = batch row). Behaviour follows models/instance_bank.py of the
reference (cited per method); the state handling is organised around one `_State` record so that it
can live either in ordinary tensors or in persistent device buffers updated in place (the form a
replayed hipGraph needs: simpb_amd/runner.py)."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import _lib
from .ops import _ptr, _stream
from .registry import PLUGIN_LAYERS, build_from_cfg

__all__ = ["InstanceBank"]

# routes.fused_bank: with the persistent (static) state the three touch points of a frame run as csrc/bank.hip
# launches (1 + 2 + 2) instead of ~70 PyTorch launches; off: the PyTorch statement below, which is also what the
# eager (non-static) mode always runs and what tests compare the kernels with.
from . import routes

_STATE_FIELDS = ("cached_feature", "cached_anchor", "confidence", "instance_id", "prev_id")


def topk(confidence, k, *inputs):
    """Rows of every input at the k largest confidences per batch row (instance_bank.py:13-20)."""
    values, index = _rank(confidence, k)
    picked = [torch.gather(t, 1, index[..., None].expand(-1, -1, t.shape[-1])) for t in inputs]
    return values, picked


def _rank(confidence, k):
    """(values, indices) of the k largest entries per row, sorted descending. On the GPU one launch of
    the bitonic row sort (csrc/rowops.hip; ties towards the lower index) instead of torch.topk's
    radix-select + radix-sort pair."""
    if confidence.is_cuda and confidence.dim() == 2 and confidence.shape[1] <= 2048:
        from .ops import topk_rows
        return topk_rows(confidence, k)
    return torch.topk(confidence, k, dim=1)


@PLUGIN_LAYERS.register_module()
class InstanceBank(nn.Module):
    def __init__(self, num_anchor, embed_dims, anchor, anchor_handler=None, num_temp_instances=0,
                 default_time_interval=0.5, confidence_decay=0.6, anchor_grad=True, feat_grad=True,
                 max_time_interval=2):
        super().__init__()
        if isinstance(anchor, str):
            table = np.load(anchor)  # the k-means anchor file of the config (:50-55)
        else:
            table = np.asarray(anchor)
        table = table[:num_anchor]
        self.num_anchor = len(table)
        self.embed_dims = embed_dims
        self.num_temp_instances = num_temp_instances
        self.default_time_interval = default_time_interval
        self.confidence_decay = confidence_decay
        self.max_time_interval = max_time_interval
        self.anchor_handler = None
        if anchor_handler is not None:
            self.anchor_handler = build_from_cfg(anchor_handler, PLUGIN_LAYERS)
            if not hasattr(self.anchor_handler, "anchor_projection"):
                raise TypeError("anchor_handler must provide anchor_projection")
        self.anchor_init = table
        self.anchor = nn.Parameter(torch.as_tensor(table, dtype=torch.float32).clone(), requires_grad=anchor_grad)
        self.instance_feature = nn.Parameter(torch.zeros(self.num_anchor, embed_dims), requires_grad=feat_grad)
        self._static = None
        self.reset()

    def init_weight(self):
        """instance_bank.py:66-69."""
        with torch.no_grad():
            self.anchor.copy_(torch.as_tensor(self.anchor_init, dtype=self.anchor.dtype))
        if self.instance_feature.requires_grad:
            nn.init.xavier_uniform_(self.instance_feature.data, gain=1)

    # ------------------------------------------------------------------ state
    def reset(self):
        """Forget the temporal state (instance_bank.py:71-79)."""
        for name in _STATE_FIELDS:
            setattr(self, name, None)
        self.prev_id = 0
        self.metas = self.mask = self.temp_confidence = self._kept_index = None
        self.has_history = False
        if self._static is not None:
            self._static["instance_id"].fill_(-1)
            self._static["prev_id"].zero_()
            self.instance_id = self._static["instance_id"]
            self.prev_id = self._static["prev_id"]

    def enable_static(self, batch_size, device):
        """Keep the temporal state in persistent device buffers that are updated in place, so a
        captured frame reads last frame's state and writes this frame's at fixed addresses.
        Semantics are unchanged; only where the tensors live."""
        t, n = self.num_temp_instances, self.num_anchor
        self._static = dict(
            cached_feature=torch.zeros(batch_size, t, self.embed_dims, device=device),
            cached_anchor=torch.zeros(batch_size, t, self.anchor.shape[-1], device=device),
            confidence=torch.zeros(batch_size, t, device=device),
            instance_id=torch.full((batch_size, n), -1, dtype=torch.long, device=device),
            prev_id=torch.zeros((), dtype=torch.long, device=device),
        )
        # two words the per-stream commit kernel (csrc/bank.hip bank_cache_streams_kernel) meets on; not part of the bank's state
        self._sync = torch.zeros(2, dtype=torch.int32, device=device)
        self.reset()

    def _keep(self, name, value):
        """Bind state `name`: plain rebinding by default, in-place copy into the persistent buffer
        in static mode."""
        if self._static is not None and name in self._static:
            self._static[name].copy_(value)
            value = self._static[name]
        setattr(self, name, value)

    # ------------------------------------------------------------------ frame start
    def _ego_motion(self, metas):
        """T_temp2cur f32 [bs, 4, 4] = inv(T_global of this frame) @ T_global of the cached frame,
        from host metadata (instance_bank.py:90-97)."""
        mats = [cur["T_global_inv"] @ old["T_global"] for cur, old in zip(metas["img_metas"], self.metas["img_metas"])]
        return torch.from_numpy(np.stack(mats).astype(np.float32))

    def _warp_cached(self, stored_anchor, T_temp2cur, dt):
        """Move the cached anchors into the current ego frame and advance them by their velocity
        (instance_bank.py:98-101, through the handler's anchor_projection with time_intervals=[-dt])."""
        if self.anchor_handler is None:
            return stored_anchor
        return self.anchor_handler.anchor_projection(stored_anchor, [T_temp2cur], time_intervals=[-dt])[0]

    def learned(self, batch_size):
        """(feature, anchor) of the learned tables alone (instance_bank.py:81-82): what the single-frame decoder layer
        starts from; touches no temporal state."""
        return self.instance_feature[None].expand(batch_size, -1, -1), self.anchor[None].expand(batch_size, -1, -1)

    def get(self, batch_size, metas=None, dn_metas=None):
        """instance_bank.py:79-119 -> (feature, anchor, cached feature, cached anchor, time step).
        The learned tables are expanded, not tiled: they are read-only downstream."""
        if dn_metas is not None:
            raise NotImplementedError("denoising anchors only exist in training")
        feature = self.instance_feature[None].expand(batch_size, -1, -1)
        anchor = self.anchor[None].expand(batch_size, -1, -1)
        static_warm = self._static is not None and self.has_history and "bank_inputs" in metas
        eager_warm = (not static_warm and self.cached_anchor is not None
                      and batch_size == self.cached_anchor.shape[0])
        if not (static_warm or eager_warm):
            self.reset()
            dt = feature.new_full((batch_size,), self.default_time_interval)
            return feature, anchor, self.cached_feature, self.cached_anchor, dt
        if static_warm and self._fusable(self._static["cached_anchor"]):
            T_temp2cur, dt = metas["bank_inputs"]
            stored = self._static["cached_anchor"]
            self.cached_feature = self._static["cached_feature"]
            bs, t = stored.shape[:2]
            warped = torch.empty_like(stored)
            self.mask = torch.empty(bs, dtype=torch.bool, device=stored.device)
            dt_out = torch.empty(bs, dtype=torch.float32, device=stored.device)
            lib = _lib.lib()
            _lib.check(lib.simpb_bank_get(_ptr(warped), _ptr(self.mask), _ptr(dt_out), _ptr(stored),
                                          _ptr(T_temp2cur.contiguous().float()), _ptr(dt.contiguous().float()), bs, t,
                                          float(self.max_time_interval), float(self.default_time_interval), _stream()),
                       "simpb_bank_get")
            self.cached_anchor = warped
            return feature, anchor, self.cached_feature, self.cached_anchor, dt_out
        if static_warm:
            # T_temp2cur and the raw time step were prepared on the host by the caller (they depend on
            # metadata only) and already sit in device buffers
            T_temp2cur, dt = metas["bank_inputs"]
            stored = self._static["cached_anchor"]
            self.cached_feature = self._static["cached_feature"]
        else:
            dt = (metas["timestamp"] - self.metas["timestamp"]).to(dtype=feature.dtype)
            T_temp2cur = self._ego_motion(metas).to(self.cached_anchor.device, non_blocking=True)
            stored = self.cached_anchor
        self.mask = torch.abs(dt) <= self.max_time_interval
        self.cached_anchor = self._warp_cached(stored, T_temp2cur, dt)
        usable = torch.logical_and(dt != 0, self.mask)
        dt = dt.masked_fill(~usable, self.default_time_interval)
        return feature, anchor, self.cached_feature, self.cached_anchor, dt

    # ------------------------------------------------------------------ after the first decoder layer
    def rank_current(self, instance_feature, confidence):
        """The ranking step of update() (instance_bank.py:137: top (num_anchor - num_temp) current instances by max-class
        logit): it depends on the first decoder layer's classification only, not on the bank, so callers that overlap frames
        run it before they wait for the previous frame (SimPBHead.forward_split). None where the fused route does not apply."""
        if not self._fusable(instance_feature) or instance_feature.shape[1] != self.num_anchor:
            return None
        bs, a, _ = instance_feature.shape
        cls = confidence.contiguous().float()
        index = torch.empty(bs, a - self.num_temp_instances, dtype=torch.int32, device=cls.device)
        _lib.check(_lib.lib().simpb_bank_update_rank(_ptr(index), _ptr(cls), bs, a, cls.shape[-1], self.num_temp_instances,
                                                     _stream()), "simpb_bank_update_rank")
        return index

    def update(self, instance_feature, anchor, confidence, rank=None, embed=None, hold=None, sticky=None):
        """Replace the 900 current instances by [cached 600 | best 300 current] for streams whose
        history is valid (instance_bank.py:121-150). rank: rank_current()'s result, if the caller took it early. embed =
        (embedding of `anchor`, embedding of the cached anchors get() returned): the merged set's embedding is then
        returned as a third value (rows follow their anchors), or None where the fused route does not apply."""
        if self.cached_feature is None:
            return (instance_feature, anchor) if embed is None else (instance_feature, anchor, embed[0])
        if instance_feature.shape[1] > self.num_anchor:
            raise NotImplementedError("denoising instances only exist in training")
        if (self._fusable(instance_feature) and self.mask is not None and self.mask.dtype == torch.bool
                and instance_feature.shape[1] == self.num_anchor):
            bs, a, c = instance_feature.shape
            t = self.num_temp_instances
            dev = instance_feature.device
            out_f = torch.empty(bs, a, c, device=dev)
            out_a = torch.empty(bs, a, anchor.shape[-1], device=dev)
            if rank is None:
                rank = self.rank_current(instance_feature, confidence)
            ids = self.instance_id if self.instance_id is self._static["instance_id"] else None
            out_e = cur_e = cached_e = None
            e_dim = 0
            if embed is not None and embed[1] is not None and embed[1].shape[1] == t and embed[0].shape[-1] % 4 == 0:
                cur_e, cached_e = embed[0].contiguous().float(), embed[1].contiguous().float()
                e_dim = cur_e.shape[-1]
                out_e = torch.empty(bs, a, e_dim, device=dev)
            _lib.check(_lib.lib().simpb_bank_update_merge(
                _ptr(out_f), _ptr(out_a), _ptr(out_e) if out_e is not None else None, _ptr(ids) if ids is not None else None,
                _ptr(rank), _ptr(instance_feature.contiguous().float()), _ptr(anchor.contiguous().float()),
                _ptr(cur_e) if cur_e is not None else None, _ptr(self.cached_feature.contiguous()),
                _ptr(self.cached_anchor.contiguous()), _ptr(cached_e) if cached_e is not None else None, _ptr(self.mask),
                _ptr(hold) if hold is not None else None, 0 if hold is None else hold.numel(),
                _ptr(sticky) if sticky is not None else None, bs, a, t, c, e_dim, _stream()),
                "simpb_bank_update_merge")
            if ids is None and self.instance_id is not None:
                self._keep("instance_id", self.instance_id.masked_fill(~self.mask[:, None], -1))
            return (out_f, out_a) if embed is None else (out_f, out_a, out_e)
        fresh = self.num_anchor - self.num_temp_instances
        _, (best_feature, best_anchor) = topk(confidence.max(dim=-1).values, fresh, instance_feature, anchor)
        merged_feature = torch.cat([self.cached_feature, best_feature], dim=1)
        merged_anchor = torch.cat([self.cached_anchor, best_anchor], dim=1)
        keep = self.mask[:, None, None]
        instance_feature = torch.where(keep, merged_feature, instance_feature)
        anchor = torch.where(keep, merged_anchor, anchor)
        if self.instance_id is not None:
            self._keep("instance_id", self.instance_id.masked_fill(~self.mask[:, None], -1))
        return (instance_feature, anchor) if embed is None else (instance_feature, anchor, None)

    # ------------------------------------------------------------------ frame end
    def cache(self, instance_feature, anchor, confidence, metas=None, feature_maps=None):
        """Keep the 600 most confident instances for the next frame; confidences of instances that
        were already tracked decay but never drop below their new score (instance_bank.py:152-167)."""
        if self.num_temp_instances <= 0:
            return
        self.metas = metas
        score = confidence.detach().max(dim=-1).values.sigmoid()
        if self.confidence is not None:
            t = self.num_temp_instances
            score[:, :t] = torch.maximum(self.confidence * self.confidence_decay, score[:, :t])
        self.temp_confidence = score
        kept_score, index = _rank(score, self.num_temp_instances)
        self._kept_index = index  # the same ranking decides which track ids survive (update_instance_id)
        pick = lambda t: torch.gather(t, 1, index[..., None].expand(-1, -1, t.shape[-1]))  # noqa: E731
        kept_feature, kept_anchor = pick(instance_feature.detach()), pick(anchor.detach())
        self._keep("confidence", kept_score)
        self._keep("cached_feature", kept_feature)
        self._keep("cached_anchor", kept_anchor)
        self.has_history = True

    def _fusable(self, t):
        return (routes.R.fused_bank and self._static is not None and t.is_cuda and self.anchor.shape[-1] == 11
                and self.num_anchor <= 1024 and 0 < self.num_temp_instances < self.num_anchor
                and self.embed_dims % 4 == 0)

    def cache_and_assign_ids(self, instance_feature, anchor, confidence, metas=None, threshold=None, hold=None, sticky=None):
        """cache() followed by get_instance_id() (simpb_head.py:744-747) on the persistent state, as two
        launches (csrc/bank.hip). Returns the instance ids, or None when the fused route does not apply
        (the caller then runs the two methods). `hold` (i32 flags on the device): when any is set the launches write
        nothing, i.e. the persistent state stays as the frame found it (an overflowed frame is re-run: runner.py)."""
        if not self._fusable(instance_feature) or instance_feature.shape[1] != self.num_anchor:
            return None
        st = self._static
        bs, a, c = instance_feature.shape
        t = self.num_temp_instances
        cls = confidence.detach().contiguous().float()
        ids_out = torch.empty(bs, a, dtype=torch.long, device=cls.device)
        scratch = torch.empty(bs, t, dtype=torch.int32, device=cls.device)
        has_prev = self.confidence is not None
        # a batch of streams: one workgroup per stream (they meet once inside the launch); a batch of one: the serial kernel
        sync = self._sync if (bs > 1 and routes.R.bank_cache_per_stream and getattr(self, "_sync", None) is not None) else None
        _lib.check(_lib.lib().simpb_bank_cache_streams(
            _ptr(st["confidence"]), _ptr(st["cached_feature"]), _ptr(st["cached_anchor"]), _ptr(st["instance_id"]),
            _ptr(st["prev_id"]), _ptr(ids_out), _ptr(scratch), _ptr(instance_feature.detach().contiguous().float()),
            _ptr(anchor.detach().contiguous().float()), _ptr(cls), bs, a, cls.shape[-1], t, c, 1 if has_prev else 0,
            float(self.confidence_decay), 0 if threshold is None else 1, 0.0 if threshold is None else float(threshold),
            _ptr(hold) if hold is not None else None, 0 if hold is None else hold.numel(),
            _ptr(sticky) if sticky is not None else None, _ptr(sync) if sync is not None else None, _stream()), "simpb_bank_cache")
        self.metas = metas
        self.confidence, self.cached_feature, self.cached_anchor = st["confidence"], st["cached_feature"], st["cached_anchor"]
        self.instance_id, self.prev_id = st["instance_id"], st["prev_id"]
        self.temp_confidence = self._kept_index = None
        self.has_history = True
        return ids_out

    def get_instance_id(self, confidence, anchor=None, threshold=None):
        """Track ids: tracked instances keep theirs, the others get fresh consecutive ids
        (instance_bank.py:169-184). Fresh ids are numbered on the device (cumsum over the mask), so
        no count is read back."""
        score = confidence.max(dim=-1).values.sigmoid()
        ids = torch.full(score.shape, -1, dtype=torch.long, device=score.device)
        if self.instance_id is not None and self.instance_id.shape[0] == ids.shape[0]:
            ids[:, : self.instance_id.shape[1]] = self.instance_id
        fresh = ids < 0
        if threshold is not None:
            fresh = fresh & (score >= threshold)
        serial = torch.cumsum(fresh.flatten().long(), 0).reshape(fresh.shape) - 1
        ids = torch.where(fresh, serial + self.prev_id, ids)
        self._keep("prev_id", self.prev_id + fresh.sum())
        self.update_instance_id(ids, score)
        return ids

    def update_instance_id(self, instance_id=None, confidence=None):
        """Ids of the instances that cache() kept, padded with -1 (instance_bank.py:186-196)."""
        if self.temp_confidence is not None and getattr(self, "_kept_index", None) is not None \
                and self._kept_index.shape[0] == instance_id.shape[0]:
            kept = torch.gather(instance_id, 1, self._kept_index)  # cache() ranked temp_confidence already
        else:
            if self.temp_confidence is not None:
                rank_by = self.temp_confidence
            else:
                rank_by = confidence.max(dim=-1).values if confidence.dim() == 3 else confidence
            kept = topk(rank_by, self.num_temp_instances, instance_id[..., None])[1][0].squeeze(dim=-1)
        self._keep("instance_id", F.pad(kept, (0, self.num_anchor - self.num_temp_instances), value=-1))
