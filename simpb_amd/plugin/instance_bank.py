"""models/instance_bank.py of the reference: learned anchors/features plus the recurrent
600-instance temporal state (one state per stream = batch row)."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .registry import PLUGIN_LAYERS, build_from_cfg

__all__ = ["InstanceBank"]


def topk(confidence, k, *inputs):
    """instance_bank.py:13-20."""
    confidence, indices = torch.topk(confidence, k, dim=1)
    outputs = [torch.gather(x, 1, indices[..., None].expand(-1, -1, x.shape[-1])) for x in inputs]
    return confidence, outputs


@PLUGIN_LAYERS.register_module()
class InstanceBank(nn.Module):
    def __init__(self, num_anchor, embed_dims, anchor, anchor_handler=None, num_temp_instances=0,
                 default_time_interval=0.5, confidence_decay=0.6, anchor_grad=True, feat_grad=True,
                 max_time_interval=2):
        super().__init__()
        self.embed_dims = embed_dims
        self.num_temp_instances = num_temp_instances
        self.default_time_interval = default_time_interval
        self.confidence_decay = confidence_decay
        self.max_time_interval = max_time_interval
        if anchor_handler is not None:
            anchor_handler = build_from_cfg(anchor_handler, PLUGIN_LAYERS)
            assert hasattr(anchor_handler, "anchor_projection")
        self.anchor_handler = anchor_handler
        if isinstance(anchor, str):
            anchor = np.load(anchor)
        elif isinstance(anchor, (list, tuple)):
            anchor = np.array(anchor)
        self.num_anchor = min(len(anchor), num_anchor)
        anchor = np.asarray(anchor)[:num_anchor]
        self.anchor_init = anchor
        self.anchor = nn.Parameter(torch.tensor(anchor, dtype=torch.float32), requires_grad=anchor_grad)
        self.instance_feature = nn.Parameter(torch.zeros([self.anchor.shape[0], self.embed_dims]),
                                             requires_grad=feat_grad)
        self._static = None
        self.reset()

    def init_weight(self):
        self.anchor.data = self.anchor.data.new_tensor(self.anchor_init)
        if self.instance_feature.requires_grad:
            torch.nn.init.xavier_uniform_(self.instance_feature.data, gain=1)

    def reset(self):
        self.cached_feature = None
        self.cached_anchor = None
        self.metas = None
        self.mask = None
        self.confidence = None
        self.temp_confidence = None
        self.instance_id = None
        self.prev_id = 0
        self.has_history = False
        if getattr(self, "_static", None) is not None:
            self._static["instance_id"].fill_(-1)
            self._static["prev_id"].zero_()
            self.instance_id = self._static["instance_id"]
            self.prev_id = self._static["prev_id"]

    # ---------------------------------------------------------------- static (graph-replayable) state
    def enable_static(self, batch_size, device):
        """Keep the temporal state in persistent device buffers that are updated in place, so a
        captured frame (hipGraph) reads last frame's state and writes this frame's at fixed
        addresses. Semantics are unchanged; only where the tensors live."""
        t, n = self.num_temp_instances, self.num_anchor
        self._static = dict(
            cached_feature=torch.zeros(batch_size, t, self.embed_dims, device=device),
            cached_anchor=torch.zeros(batch_size, t, self.anchor.shape[-1], device=device),
            confidence=torch.zeros(batch_size, t, device=device),
            instance_id=torch.full((batch_size, n), -1, dtype=torch.long, device=device),
            prev_id=torch.zeros((), dtype=torch.long, device=device),
        )
        self.reset()

    def _keep(self, name, value):
        """Bind state `name`: rebinding in the default mode, in-place copy into the persistent
        buffer in static mode."""
        st = getattr(self, "_static", None)
        if st is not None and name in st:
            st[name].copy_(value)
            value = st[name]
        setattr(self, name, value)

    def get(self, batch_size, metas=None, dn_metas=None):
        """instance_bank.py:79-119. `expand` instead of `tile`: the learned tables are read-only
        downstream, so no [bs, 900, 256] copy is made."""
        instance_feature = self.instance_feature[None].expand(batch_size, -1, -1)
        anchor = self.anchor[None].expand(batch_size, -1, -1)
        if self._static is not None and self.has_history and "bank_inputs" in metas:
            # static mode: T_temp2cur f32[bs,4,4] and the raw time step f32[bs] were prepared by the
            # caller on the host (they only depend on metadata) and already sit in device buffers
            T_temp2cur, time_interval = metas["bank_inputs"]
            self.mask = torch.abs(time_interval) <= self.max_time_interval
            self.cached_feature = self._static["cached_feature"]
            self.cached_anchor = self.anchor_handler.anchor_projection(
                self._static["cached_anchor"], [T_temp2cur], time_intervals=[-time_interval])[0]
            time_interval = time_interval.masked_fill(
                ~torch.logical_and(time_interval != 0, self.mask), self.default_time_interval)
        elif self.cached_anchor is not None and batch_size == self.cached_anchor.shape[0]:
            history_time = self.metas["timestamp"]
            time_interval = (metas["timestamp"] - history_time).to(dtype=instance_feature.dtype)
            self.mask = torch.abs(time_interval) <= self.max_time_interval
            if self.anchor_handler is not None:
                T_temp2cur = np.stack([x["T_global_inv"] @ self.metas["img_metas"][i]["T_global"]
                                       for i, x in enumerate(metas["img_metas"])])
                T_temp2cur = torch.from_numpy(T_temp2cur.astype(np.float32)).to(self.cached_anchor.device,
                                                                              non_blocking=True)
                self.cached_anchor = self.anchor_handler.anchor_projection(
                    self.cached_anchor, [T_temp2cur], time_intervals=[-time_interval])[0]
            if dn_metas is not None:
                raise NotImplementedError("denoising anchors only exist in training")
            time_interval = time_interval.masked_fill(
                ~torch.logical_and(time_interval != 0, self.mask), self.default_time_interval)
        else:
            self.reset()
            time_interval = instance_feature.new_tensor([self.default_time_interval] * batch_size)
        return instance_feature, anchor, self.cached_feature, self.cached_anchor, time_interval

    def update(self, instance_feature, anchor, confidence):
        """instance_bank.py:121-150."""
        if self.cached_feature is None:
            return instance_feature, anchor
        if instance_feature.shape[1] > self.num_anchor:
            raise NotImplementedError("denoising instances only exist in training")
        N = self.num_anchor - self.num_temp_instances
        confidence = confidence.max(dim=-1).values
        _, (selected_feature, selected_anchor) = topk(confidence, N, instance_feature, anchor)
        selected_feature = torch.cat([self.cached_feature, selected_feature], dim=1)
        selected_anchor = torch.cat([self.cached_anchor, selected_anchor], dim=1)
        instance_feature = torch.where(self.mask[:, None, None], selected_feature, instance_feature)
        anchor = torch.where(self.mask[:, None, None], selected_anchor, anchor)
        if self.instance_id is not None:
            self._keep("instance_id", self.instance_id.masked_fill(~self.mask[:, None], -1))
        return instance_feature, anchor

    def cache(self, instance_feature, anchor, confidence, metas=None, feature_maps=None):
        """instance_bank.py:152-167."""
        if self.num_temp_instances <= 0:
            return
        instance_feature = instance_feature.detach()
        anchor = anchor.detach()
        confidence = confidence.detach()
        self.metas = metas
        confidence = confidence.max(dim=-1).values.sigmoid()
        if self.confidence is not None:
            confidence[:, : self.num_temp_instances] = torch.maximum(
                self.confidence * self.confidence_decay, confidence[:, : self.num_temp_instances])
        self.temp_confidence = confidence
        conf, (feat, anc) = topk(confidence, self.num_temp_instances, instance_feature, anchor)
        self._keep("confidence", conf)
        self._keep("cached_feature", feat)
        self._keep("cached_anchor", anc)
        self.has_history = True

    def get_instance_id(self, confidence, anchor=None, threshold=None):
        """instance_bank.py:169-184; new ids are numbered on the device (cumsum over the mask) so
        the count is only read back to advance prev_id."""
        confidence = confidence.max(dim=-1).values.sigmoid()
        instance_id = confidence.new_full(confidence.shape, -1).long()
        if self.instance_id is not None and self.instance_id.shape[0] == instance_id.shape[0]:
            instance_id[:, : self.instance_id.shape[1]] = self.instance_id
        mask = instance_id < 0
        if threshold is not None:
            mask = mask & (confidence >= threshold)
        order = torch.cumsum(mask.flatten().long(), 0).reshape(mask.shape) - 1
        instance_id = torch.where(mask, order + self.prev_id, instance_id)
        self._keep("prev_id", self.prev_id + mask.sum())
        self.update_instance_id(instance_id, confidence)
        return instance_id

    def update_instance_id(self, instance_id=None, confidence=None):
        """instance_bank.py:186-196."""
        if self.temp_confidence is None:
            temp_conf = confidence.max(dim=-1).values if confidence.dim() == 3 else confidence
        else:
            temp_conf = self.temp_confidence
        instance_id = topk(temp_conf, self.num_temp_instances, instance_id[..., None])[1][0].squeeze(dim=-1)
        self._keep("instance_id", F.pad(instance_id, (0, self.num_anchor - self.num_temp_instances), value=-1))
