"""models/simpb_head.py of the reference: SimPBHead, inference path (forward :323-747 and
post_process :1089-1123). The 50-entry operation_order interpreter, the module names and the
parameter names are the reference's; the data movement between operators uses index tables from
the allocation kernels instead of one-hot matrices."""
from typing import List, Optional

import torch
import torch.nn as nn

from . import training_stubs  # noqa: F401  (registers the inert training-only classes)
from .allocation import gather_rows
from .registry import (ATTENTION, BBOX_CODERS, BBOX_SAMPLERS, FEEDFORWARD_NETWORK, HEADS, LOSSES, NORM_LAYERS,
                       PLUGIN_LAYERS, POSITIONAL_ENCODING, TRANSFORMER_LAYER_SEQUENCE, build_from_cfg)
from . import dense, routes
from .layers import BaseModule, fused_graph_attention

__all__ = ["SimPBHead"]


@HEADS.register_module()
class SimPBHead(BaseModule):
    def __init__(self, instance_bank: dict, anchor_encoder: dict, graph_model: dict, norm_layer: dict, ffn: dict,
                 deformable_model: dict, num_cams: int = 6, num_decoder: int = 6, num_single_frame_decoder: int = -1,
                 temp_graph_model: dict = None, loss_cls: dict = None, loss_reg: dict = None, decoder: dict = None,
                 sampler: dict = None, reg_weights: List = None, operation_order: Optional[List[str]] = None,
                 cls_threshold_to_reg: float = -1, dn_loss_weight: float = 5.0, decouple_attn: bool = True,
                 init_cfg: dict = None, enable2d=False, enable3d=True, embed_dims=256, num_levels=4, num_anchor=900,
                 encoder2d=None, share_encoder2d=False, anchor_encoder2d=None, positional_encoding=None,
                 qg_self_attn=None, qg_cross_attn=None, refine_layer2d=None, refine_layer3d=None,
                 decouple_attn2d=False, with_allocate_attn_mask=False, dynamic_allocation=None,
                 adaptive_aggregation=None, coster2d=None, coster3d=None, denoise2d=None, loss_cls2d=None,
                 loss_iou2d=None, loss_bbox2d=None, loss_alpha2d=None, loss_depth2d=None, **kwargs):
        super().__init__(init_cfg)
        if encoder2d is not None:
            raise NotImplementedError("the 2D encoder variant ('SimPB' non-plus) has no released config "
                                      "(SURVEY.md §8f item 4)")
        if not (decouple_attn and enable3d):
            raise NotImplementedError("the SimPB configs use decouple_attn=True, enable3d=True")
        self.enable2d = enable2d
        self.enable3d = enable3d
        self.embed_dims = embed_dims
        self.num_cams = num_cams
        self.num_levels = num_levels
        self.num_anchor = num_anchor
        self.num_decoder = num_decoder
        self.num_single_frame_decoder = num_single_frame_decoder
        self.dn_loss_weight = dn_loss_weight
        self.decouple_attn = decouple_attn
        self.decouple_attn2d = decouple_attn2d
        self.cls_threshold_to_reg = cls_threshold_to_reg
        self.reg_weights = [1.0] * 10 if reg_weights is None else reg_weights
        if operation_order is None:
            operation_order = ["temp_gnn", "gnn", "norm", "deformable", "norm", "ffn", "norm", "refine3d"] * num_decoder
            operation_order = operation_order[3:]
        self.operation_order = list(operation_order)

        def build(cfg, registry):
            return None if cfg is None else build_from_cfg(cfg, registry)

        self.with_denoise2d = False
        if self.enable2d:
            self.encoder2d = None
            if anchor_encoder2d is None:
                raise NotImplementedError("the SimPB configs always give anchor_encoder2d")
            self.anchor_encoder2d = build(anchor_encoder2d, POSITIONAL_ENCODING)
            self.instance_status = "3d"
            self.share_encoder2d = share_encoder2d
            self.with_allocate_attn_mask = with_allocate_attn_mask
            self.loss_cls2d = build(loss_cls2d, LOSSES)
            self.loss_iou2d = build(loss_iou2d, LOSSES)
            self.loss_bbox2d = build(loss_bbox2d, LOSSES)
            self.loss_alpha2d = build(loss_alpha2d, LOSSES)
            self.loss_depth2d = build(loss_depth2d, LOSSES)
            if denoise2d is not None:
                self.with_denoise2d = True
                self.denoise2d = build(denoise2d, PLUGIN_LAYERS)
            self.coster2d = build(coster2d, BBOX_SAMPLERS)
        self.instance_bank = build(instance_bank, PLUGIN_LAYERS)
        self.anchor_encoder = build(anchor_encoder, POSITIONAL_ENCODING)
        self.sampler = build(sampler, BBOX_SAMPLERS)
        self.decoder = build(decoder, BBOX_CODERS)
        self.loss_cls = build(loss_cls, LOSSES)
        self.loss_reg = build(loss_reg, LOSSES)
        self.op_config_map = {
            "ffn": [ffn, FEEDFORWARD_NETWORK],
            "norm": [norm_layer, NORM_LAYERS],
            "allocation": [dynamic_allocation, PLUGIN_LAYERS],
            "aggregation": [adaptive_aggregation, PLUGIN_LAYERS],
            "qg_self_attn": [qg_self_attn, ATTENTION],
            "qg_cross_attn": [qg_cross_attn, ATTENTION],
            "refine2d": [refine_layer2d, PLUGIN_LAYERS],
            "gnn": [graph_model, ATTENTION],
            "temp_gnn": [temp_graph_model, ATTENTION],
            "deformable": [deformable_model, ATTENTION],
            "refine3d": [refine_layer3d, PLUGIN_LAYERS],
        }
        self.layers = nn.ModuleList([build(*self.op_config_map.get(op, [None, None])) for op in self.operation_order])
        self.fc_before = nn.Linear(self.embed_dims, self.embed_dims * 2, bias=False)
        self.fc_after = nn.Linear(self.embed_dims * 2, self.embed_dims, bias=False)
        if self.decouple_attn2d and self.enable2d:
            self.fc_before2d = nn.Linear(self.embed_dims, self.embed_dims * 2, bias=False)
            self.fc_after2d = nn.Linear(self.embed_dims * 2, self.embed_dims, bias=False)
        else:
            self.fc_before2d = nn.Identity()
            self.fc_after2d = nn.Identity()
        self.use_deformable_func = True  # set by SimPB.__init__ in the reference (simpb.py:53)
        self._tables = None
        self._m_live = None  # device int: live 2D slots of the static slot array (dense.py), 2D state only
        # None: size the 2D query set exactly each frame (one count readback, the reference's
        # behaviour). An int: static shapes with that many 2D slots and no host round trip inside
        # the frame, which is what lets simpb_amd.runner replay the frame as one hipGraph.
        self.static_capacity = None
        # True (static capacity only): the batch is a set of INDEPENDENT camera streams -- every stream keeps the 2D query
        # set a batch of one gives it, laid out as one flat slot array (allocation.allocate_independent),
        # instead of the reference's groups padded to the max over the batch (allocation.py:91-99). False: the reference's
        # batch semantics.
        self.independent_streams = False

    def init_weights(self):
        """simpb_head.py:202-212."""
        for i, op in enumerate(self.operation_order):
            if self.layers[i] is None:
                continue
            for p in self.layers[i].parameters():
                if p.dim() > 1:
                    nn.init.xavier_uniform_(p)
        for m in self.modules():
            if hasattr(m, "init_weight"):
                m.init_weight()

    # ------------------------------------------------------------------ feature-map tables
    def prepare2d(self, feature_maps, metas):
        """simpb_head.py:281-292 (encoder-less branch): the channel-last token buffer viewed per
        camera, plus i32/i64 copies of the (H, W, start) tables cached across frames."""
        if not self.use_deformable_func:
            raise RuntimeError("SimPBHead needs use_deformable_func=True (simpb_head.py:293-294)")
        col, spatial_shape, scale_start = feature_maps[:3]
        bs, _, dim = col.shape
        nc = len(spatial_shape)
        key = (spatial_shape.data_ptr(), scale_start.data_ptr(), col.device)
        if self._tables is None or self._tables[0] != key:
            self._tables = (key, spatial_shape.int().contiguous(), scale_start.int().contiguous(),
                            spatial_shape[0].long().contiguous(), scale_start[0].long().contiguous())
        _, ss32, st32, ss_cam, st_cam = self._tables
        feat_flatten = col.reshape(bs, nc, -1, dim).flatten(0, 1)
        half = getattr(col, "simpb_f16", None)   # the same tokens as the fp16 backbone left them (detector.FPN)
        encoder2d_dict = {
            "value_f16": half.reshape(bs, nc, -1, dim).flatten(0, 1) if half is not None and half.shape == col.shape else None,
            "value": feat_flatten,
            "key_padding_mask": None,  # all-False in the reference (:286): masked_fill would be a no-op
            "spatial_shapes": ss_cam,
            "level_start_index": st_cam,
        }
        return encoder2d_dict, [col, ss32, st32]

    def precompute_values(self, feature_maps):
        """value_proj of every qg_cross_attn layer (group_attn.py:176) applied to the token buffer.
        It depends on nothing but the features and the weights, so a caller that overlaps the
        backbone of the next frame with this frame's decoder (runner.PipelinedRunner) runs it with the
        backbone, off the decoder's critical path. Returns {layer index: projected tokens}."""
        col = feature_maps[0]
        out = {}
        if routes.R.msda_linear and routes.R.dense:
            return out   # value_proj is applied behind the sampling (group_attn._forward_linear): nothing to precompute
        for i, op in enumerate(self.operation_order):
            if op == "qg_cross_attn":
                out[i] = self.layers[i].project_value(col)
        return out

    # ------------------------------------------------------------------ decoupled attention
    def graph_model(self, index, query, key=None, value=None, query_pos=None, key_pos=None, **kwargs):
        """simpb_head.py:298-310."""
        layer = self.layers[index] if isinstance(index, int) else index
        kwargs.pop("attn_mask", None)
        if query.is_cuda and routes.R.dense:
            out = fused_graph_attention(layer, self.fc_before, self.fc_after, query, query_pos, key, key_pos, value)
            if out is not None:
                return out
        query = torch.cat([query, query_pos], dim=-1)
        key = torch.cat([key, key_pos], dim=-1) if key is not None else None
        # fc_before (:303) is handed to the attention operator, which applies it on the value branch,
        # forked from the q/k projection branch (layers.run_parallel)
        pre = self.fc_before if value is not None else None
        return self.fc_after(layer(query, key, value, query_pos=None, key_pos=None, value_pre=pre, **kwargs))

    def graph_model2d(self, index, query, key=None, value=None, query_pos=None, key_pos=None, **kwargs):
        """simpb_head.py:312-321."""
        if self.decouple_attn2d and query.is_cuda and routes.R.dense and kwargs.get("query_cam") is not None:
            out = fused_graph_attention(self.layers[index], self.fc_before2d, self.fc_after2d, query, query_pos, key,
                                        key_pos, value, query_cam=kwargs["query_cam"],
                                        group_start=kwargs.get("group_start"), m_live=self._m_live)
            if out is not None:
                return out
        if self.decouple_attn2d:
            query = torch.cat([query, query_pos], dim=-1)
            key = torch.cat([key, key_pos], dim=-1) if key is not None else None
            query_pos, key_pos = None, None
        pre = self.fc_before2d if value is not None and isinstance(self.fc_before2d, nn.Linear) else None
        return self.fc_after2d(self.layers[index](query, key, value, query_pos=query_pos, key_pos=key_pos,
                                                  value_pre=pre, **kwargs))

    def _camera_embeddings(self, metas, batch_size):
        """camera_encoder of every deformable layer (blocks.py:93-99,174-176) in ONE chain launch at the
        start of the frame: they depend on projection_mat and their own weights only, and each is a
        6-row, latency-bound chain (22 us) that would otherwise sit on the critical path of its layer."""
        proj = metas.get("projection_mat")
        idx = [i for i, op in enumerate(self.operation_order)
               if op == "deformable" and getattr(self.layers[i], "camera_encoder", None) is not None]
        if proj is None or not proj.is_cuda or not idx or len(idx) > 8 or not routes.R.dense:
            return {}
        from . import fused
        cam_in = proj[:, :, :3].reshape(batch_size * self.num_cams, -1).float().contiguous()
        jobs, outs = [], {}
        for i in idx:
            plan = fused.plan_of(self.layers[i].camera_encoder)
            if plan.in_dim != cam_in.shape[1]:
                return {}
            out = torch.empty(cam_in.shape[0], plan.out_dim, device=cam_in.device)
            jobs.append(dict(plan=plan, x=(cam_in, cam_in.shape[1], 0), out=(out, plan.out_dim, 0)))
            outs[i] = out.reshape(batch_size, self.num_cams, plan.out_dim)
        fused.run_chains(jobs, cam_in.shape[0], cam_in.device)
        return outs

    def _next_is_ffn(self, i):
        """True when op i+1 is the FFN, whose pre-norm reads a residual_mode="cat" result as two
        segments in place (dense.Segments) instead of a materialised concatenation."""
        return i + 1 < len(self.operation_order) and self.operation_order[i + 1] == "ffn"

    # ------------------------------------------------------------------ forward
    def forward(self, feature_maps, metas: dict):
        """simpb_head.py:323-747 (inference)."""
        gen = self._forward(feature_maps, metas, split=False)
        try:
            next(gen)
        except StopIteration as done:
            return done.value
        raise RuntimeError("unsplit forward does not pause")

    def forward_split(self, feature_maps, metas: dict):
        """The same forward as a two-step generator, for callers that overlap frames (runner.SplitPipelinedRunner):
        `next(gen)` runs the single-frame decoder layer(s) -- everything up to the first InstanceBank.update
        (simpb_head.py:690-696), which reads nothing the previous frame's decoder wrote (learned anchors only; the time
        step comes in as metas["time_interval"]) -- and pauses; `gen.send(None)` runs the temporal rest (bank.get,
        update, the remaining layers, cache) and ends with StopIteration carrying the output dict. Same launches, same
        numbers as forward() except that the two anchor sets are embedded by two encoder launches instead of one."""
        return self._forward(feature_maps, metas, split=True)

    def _forward(self, feature_maps, metas: dict, split: bool):
        if self.training:
            raise NotImplementedError("SimPBHead here is the inference path; training (denoising, losses) is out of scope")
        if isinstance(feature_maps, torch.Tensor):
            feature_maps = [feature_maps]
        batch_size = feature_maps[0].shape[0]
        if self.sampler is not None and self.sampler.dn_metas is not None:
            self.sampler.dn_metas = None  # :333-334; never set in eval

        if split:
            if "time_interval" not in metas or self.static_capacity is None:
                raise ValueError("forward_split needs metas['time_interval'] (f32 [bs]) and a static capacity")
            instance_feature, anchor = self.instance_bank.learned(batch_size)
            temp_instance_feature = temp_anchor = None
            time_interval = metas["time_interval"]
        else:
            instance_feature, anchor, temp_instance_feature, temp_anchor, time_interval = self.instance_bank.get(
                batch_size, metas, dn_metas=None)
        if temp_anchor is not None and anchor.is_cuda:
            # one encoder launch over both anchor sets (the chain kernel is latency-bound per launch)
            both = self.anchor_encoder.forward(torch.cat([anchor, temp_anchor], dim=1))
            anchor_embed = dense.report(self.anchor_encoder, both[:, : anchor.shape[1]])
            temp_anchor_embed = dense.report(self.anchor_encoder, both[:, anchor.shape[1]:])
        else:
            anchor_embed = self.anchor_encoder(anchor)
            temp_anchor_embed = self.anchor_encoder(temp_anchor) if temp_anchor is not None else None

        quality, prediction, classification = [], [], []
        prediction2d, classification2d, prediction_alpha2d, prediction_depth2d = [], [], [], []
        ref_pts2d_list, ref_trans_shape_list, ref_trans_matrix_list, ref_query_groups_list = [], [], [], []
        alloc_list = []
        cap = self.static_capacity
        temp_attn_instance = instance_feature
        pre_values = feature_maps[3] if len(feature_maps) > 3 else None
        encoder2d_dict, feature_maps = self.prepare2d(feature_maps, metas)
        cam_embeds = self._camera_embeddings(metas, batch_size)
        alloc = None
        self._m_live = None
        last = len(self.operation_order) - 1
        # static mode: the overflow flags of the frame's allocation layers in one tensor; the frame-end commit of the
        # bank holds back when any is set, so that the caller can re-run the frame (runner.py) on untouched state
        pending_norm = None
        overflow = hold = sticky = None
        if cap is not None:
            n_alloc = sum(op == "allocation" for op in self.operation_order)
            chain = metas.get("overflow_chain")
            sticky = None
            if split:
                # (hb i32 [n_alloc + 1] owned by the caller, sticky i32 [1]): this frame's flags live in hb[:n_alloc] (the
                # last entry is spare); `sticky` is the word the frame decoded before this one left (1 = it held its commit
                # back): the bank kernels of the temporal part treat it like a flag of this frame, and this frame's commit
                # leaves its own verdict in it in turn (csrc/bank.hip)
                hb, sticky = metas["overflow_split"]
                if hb.dtype != torch.int32 or hb.numel() != n_alloc + 1 or sticky.dtype != torch.int32 or sticky.numel() != 1:
                    raise ValueError("overflow_split: (i32 [allocation layers + 1], i32 [1])")
                overflow = hb[:n_alloc]
                overflow.zero_()
                hold = hb
            elif chain is not None:
                # (flags i32 [rows, n_alloc] owned by the caller, this frame's row): the commit also holds back when a flag
                # of the OTHER rows is set, i.e. when the frame decoded just before this one (enqueued while its own flags
                # had not reached the host yet: runner.PipelinedRunner) is going to be re-run
                flags, row = chain
                if flags.dtype != torch.int32 or flags.dim() != 2 or flags.shape[1] != n_alloc or not flags.is_contiguous():
                    raise ValueError("overflow_chain: (contiguous i32 [rows, allocation layers] tensor, row)")
                overflow = flags[row]
                overflow.zero_()
                hold = flags.view(-1)
            else:
                overflow = hold = torch.zeros(n_alloc, dtype=torch.int32, device=anchor.device)

        for i, op in enumerate(self.operation_order):
            layer = self.layers[i]
            if layer is None:
                continue
            elif op == "norm":
                nxt = self.operation_order[i + 1] if i < last else None
                if (instance_feature.is_cuda and routes.R.dense and routes.R.norm_in_refine and routes.R.chain_rows4
                        and nxt in ("refine2d", "refine3d") and self.layers[i + 1] is not None
                        and torch.is_tensor(instance_feature) and abs(layer.eps - 1e-5) < 1e-12):
                    pending_norm = layer   # applied inside the refinement head's launch, which writes this operator's output
                elif instance_feature.is_cuda and routes.R.dense:
                    instance_feature = dense.layernorm(instance_feature, layer, m_live=self._m_live)
                else:
                    instance_feature = layer(instance_feature)
            elif op == "ffn":
                instance_feature = layer(instance_feature, m_live=self._m_live)
            elif op == "allocation":
                assert self.instance_status == "3d"
                k = len(ref_pts2d_list)
                if self.independent_streams and batch_size > 1 and cap is None:
                    raise ValueError("independent_streams needs a static capacity (the flat slot array has a fixed size)")
                ragged = self.independent_streams and batch_size > 1
                anchor2d, ref_depth2d, ref_trans_mask, ref_trans_shape, _, _, ref_query_groups, _ = layer(
                    anchor, metas, dense=False, capacity=cap,
                    overflow_out=overflow[k:k + 1] if overflow is not None else None, independent=ragged)
                alloc = layer.last
                if ragged:   # one flat slot array over batch_size * num_cams groups: the 2D operators run as a batch of one
                    groups = batch_size * self.num_cams
                    self._m_live = alloc.group_start[groups: groups + 1]
                    instance_feature = instance_feature.reshape(1, -1, instance_feature.shape[-1])
                elif cap is not None and batch_size == 1 and alloc.group_start is not None:
                    self._m_live = alloc.group_start[self.num_cams: self.num_cams + 1]
                instance_feature = gather_rows(instance_feature, alloc.q2a)  # :438
                anchor_embed2d = (self.anchor_encoder2d(anchor2d, m_live=self._m_live) if self._m_live is not None
                                  else self.anchor_encoder2d(anchor2d))
                ref_pts2d_list.append(anchor2d[..., :2])
                self.instance_status = "2d"
            elif op == "aggregation":
                assert self.instance_status == "2d"
                instance_feature, anchor_embed, anchor = layer(
                    query2d=instance_feature, query_pos2d=anchor_embed2d, anchor2d=anchor2d,
                    query3d=temp_attn_instance, query_pos3d=anchor_embed, anchor3d=anchor,
                    allocation=alloc, attn_mask=None, graph_model=self.graph_model, m_live=self._m_live)
                self.instance_status = "3d"
                self._m_live = None
            elif op == "qg_self_attn":
                instance_feature = self.graph_model2d(i, query=instance_feature, value=instance_feature,
                                                      query_pos=anchor_embed2d, query_groups=ref_query_groups,
                                                      query_cam=alloc.query_cam, group_start=alloc.group_start)
            elif op == "qg_cross_attn":
                enc = encoder2d_dict
                if pre_values is not None and i in pre_values:
                    enc = dict(encoder2d_dict, value=pre_values[i], value_is_projected=True)
                instance_feature = layer(query=instance_feature, query_pos=anchor_embed2d,
                                         reference_points=anchor2d.unsqueeze(2), query_groups=ref_query_groups,
                                         query_cam=alloc.query_cam, m_live=self._m_live,
                                         keep_parts=routes.R.dense and self._next_is_ffn(i), **enc)
            elif op == "refine2d":
                kw = dict(m_live=self._m_live) if self._m_live is not None else {}
                if pending_norm is not None:
                    kw["norm"] = pending_norm
                anchor2d, cls2d, depth2d, alpha2d = layer(instance_feature, anchor2d, anchor_embed2d, metas=metas,
                                                          query_groups=ref_query_groups, **kw)
                if pending_norm is not None:
                    instance_feature, pending_norm = dense.report(pending_norm, layer.norm_out), None
                prediction2d.append(anchor2d)
                classification2d.append(cls2d)
                prediction_alpha2d.append(alpha2d)
                prediction_depth2d.append(depth2d)
                ref_trans_shape_list.append(ref_trans_shape)
                ref_trans_matrix_list.append(alloc.q2a)  # index form of ref_trans_matrix
                ref_query_groups_list.append(ref_query_groups)
                alloc_list.append(alloc)
            elif op == "gnn":
                instance_feature = self.graph_model(i, instance_feature, value=instance_feature, query_pos=anchor_embed)
            elif op == "temp_gnn":
                instance_feature = self.graph_model(i, instance_feature, temp_instance_feature, temp_instance_feature,
                                                    query_pos=anchor_embed, key_pos=temp_anchor_embed)
                temp_attn_instance = instance_feature
            elif op == "deformable":
                instance_feature = layer(instance_feature, anchor, anchor_embed, feature_maps, metas,
                                         keep_parts=routes.R.dense and self._next_is_ffn(i),
                                         cam_embed=cam_embeds.get(i))
            elif op == "refine3d":
                kw = dict(norm=pending_norm) if pending_norm is not None else {}
                anchor, cls, qt = layer(
                    instance_feature, anchor, anchor_embed, time_interval=time_interval,
                    return_cls=(len(prediction) == self.num_single_frame_decoder - 1 or i == last), **kw)
                if pending_norm is not None:
                    instance_feature, pending_norm = dense.report(pending_norm, layer.norm_out), None
                prediction.append(anchor)
                classification.append(cls)
                quality.append(qt)
                merged_embed = None
                if len(prediction) == self.num_single_frame_decoder:
                    # what InstanceBank.update needs from THIS frame does not depend on the bank: the ranking of the current
                    # instances (by this layer's classification) and the embedding of their refined anchors. Both are taken
                    # here, in front of the point where a caller overlapping frames starts to wait for the previous frame; the
                    # update then merges embeddings along with features and anchors (rows follow their anchors), and the
                    # encoder launch behind it (:621-622) is not needed.
                    rank = cur_embed = None
                    if anchor.is_cuda and routes.R.dense and i != last:
                        rank = self.instance_bank.rank_current(instance_feature, cls)
                        if rank is not None:
                            cur_embed = self.anchor_encoder.forward(anchor)   # (no call site of the reference: no hooks)
                    if split:
                        yield "single-frame layers done"
                        # ---- the temporal part: needs what the previous frame's decoder committed
                        _, _, temp_instance_feature, temp_anchor, time_interval = self.instance_bank.get(
                            batch_size, metas, dn_metas=None)
                        temp_anchor_embed = self.anchor_encoder(temp_anchor) if temp_anchor is not None else None
                    if rank is not None:
                        instance_feature, anchor, merged_embed = self.instance_bank.update(
                            instance_feature, anchor, cls, rank=rank, embed=(cur_embed, temp_anchor_embed), hold=hold, sticky=sticky)
                    else:
                        instance_feature, anchor = self.instance_bank.update(instance_feature, anchor, cls)
                if merged_embed is not None:
                    anchor_embed = dense.report(self.anchor_encoder, merged_embed)
                elif i != last:
                    anchor_embed = self.anchor_encoder(anchor)
                if len(prediction) > self.num_single_frame_decoder and temp_anchor_embed is not None:
                    temp_anchor_embed = anchor_embed[:, : self.instance_bank.num_temp_instances]
            else:
                raise NotImplementedError(f"{op} is not supported.")

        output = {
            "quality": quality, "prediction": prediction, "classification": classification,
            "prediction2d": prediction2d, "classification2d": classification2d,
            "prediction_alpha2d": prediction_alpha2d, "prediction_depth2d": prediction_depth2d,
            "ref_pts2d_list": ref_pts2d_list, "ref_trans_shape_list": ref_trans_shape_list,
            "ref_trans_matrix_list": ref_trans_matrix_list, "ref_query_groups_list": ref_query_groups_list,
            "alloc_list": alloc_list, "overflow": overflow,
        }
        ids = self.instance_bank.cache_and_assign_ids(instance_feature, anchor, cls, metas, self.decoder.score_threshold,
                                                      hold=hold, sticky=sticky)
        if ids is None:
            self.instance_bank.cache(instance_feature, anchor, cls, metas, feature_maps)
            ids = self.instance_bank.get_instance_id(cls, anchor, self.decoder.score_threshold)
        if split and ids is None:   # (the commit kernel leaves the word itself; this is the route without it)
            torch.amax(torch.cat([hold, sticky]), dim=0, keepdim=True, out=sticky)   # what the next frame's commit has to respect
        output["instance_id"] = ids
        return output

    def loss(self, model_outs, data):
        raise NotImplementedError("training losses (simpb_head.py:749-1086) are out of scope of this path")

    def post_process(self, model_outs, data, output_idx=-1, output_idx2d=-1):
        """simpb_head.py:1089-1123."""
        results = [dict() for _ in data["img_metas"]]
        if self.enable2d:
            aug_configs = [m["aug_config"] for m in data["img_metas"]]
            results_3d = self.decoder.decode_with2d(
                model_outs["classification"], model_outs["prediction"], model_outs.get("instance_id"),
                model_outs.get("quality"), output_idx, model_outs["classification2d"], model_outs["prediction2d"],
                model_outs["ref_trans_matrix_list"], model_outs["ref_query_groups_list"], output_idx2d, aug_configs,
                with_association=True)
        else:
            results_3d = self.decoder.decode(model_outs["classification"], model_outs["prediction"],
                                             model_outs.get("instance_id"), model_outs.get("quality"),
                                             output_idx=output_idx)
        for i, result_3d in enumerate(results_3d):
            results[i]["img_bbox"] = result_3d
        return results
