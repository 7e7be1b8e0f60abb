"""Programmatic equivalents of the reference's model configs (projects/configs/
simpb_nus_r50_{img,uimg}_704x256.py :36-312) so that tests and bench.py on the GPU box, where
/root/reference does not exist, build exactly the model those files describe.
tests/test_config_parity.py compares this against the reference file itself when it is present.
`simpb_plus(depth=101, input_shape=(1408, 512))` is the derived R101 configuration of
BASELINE.json config #4 (the reference ships no such file: SURVEY.md §0)."""

CLASS_NAMES = ["car", "truck", "construction_vehicle", "bus", "trailer", "barrier", "motorcycle", "bicycle",
               "pedestrian", "traffic_cone"]

LAYER3D_FIRST = ["gnn", "norm", "deformable", "ffn", "norm", "refine3d"]
LAYER2D_FIRST = ["allocation", "qg_self_attn", "norm", "qg_cross_attn", "ffn", "norm", "refine2d", "aggregation",
                 "refine3d"]
LAYER3D = ["temp_gnn"] + LAYER3D_FIRST
LAYER2D = ["temp_gnn"] + LAYER2D_FIRST


def operation_order():
    return LAYER2D_FIRST + LAYER3D + LAYER2D + LAYER3D + LAYER2D + LAYER3D


def _mha(dims, groups, drop):
    return dict(type="MultiheadAttention", embed_dims=dims, num_heads=groups, batch_first=True, dropout=drop)


def simpb_plus(depth=50, input_shape=(704, 256), anchor="./data/nuscenes/nuscenes_kmeans900.npy", temporal=True,
               pretrained="ckpts/resnet50-19c8e357.pth"):
    e, g, lv, drop, ncls, ndn, ntdn = 256, 8, 4, 0.1, len(CLASS_NAMES), 5, 3
    cone = CLASS_NAMES.index("traffic_cone")
    head = dict(
        type="SimPBHead", enable2d=True, num_levels=lv, embed_dims=e, cls_threshold_to_reg=0.05, decouple_attn=True,
        decouple_attn2d=True, with_denoise2d=True,
        denoise2d=dict(type="Denoise2D", num_dn_groups=ndn),
        instance_bank=dict(type="InstanceBank", num_anchor=900, embed_dims=e, anchor=anchor,
                           anchor_handler=dict(type="SparseBox3DKeyPointsGenerator"),
                           num_temp_instances=600 if temporal else -1, confidence_decay=0.6, feat_grad=False),
        anchor_encoder2d=dict(type="SparseBox2DEncoder", embed_dims=e, with_sin_embed=True, in_loops=1, out_loops=2),
        anchor_encoder=dict(type="SparseBox3DEncoder", vel_dims=3, embed_dims=[128, 32, 32, 64], mode="cat",
                            output_fc=False, in_loops=1, out_loops=4),
        encoder2d=None, num_single_frame_decoder=1, operation_order=operation_order(),
        norm_layer=dict(type="LN", normalized_shape=e),
        ffn=dict(type="AsymmetricFFN", in_channels=e * 2, pre_norm=dict(type="LN"), embed_dims=e,
                 feedforward_channels=e * 4, num_fcs=2, ffn_drop=drop, act_cfg=dict(type="ReLU", inplace=True)),
        dynamic_allocation=dict(type="DynamicQueryAllocation", limit_corners_num=[100] * 6),
        adaptive_aggregation=dict(type="AdaptiveQueryAggregation", self_attn=_mha(e * 2, g, drop), reweight=True,
                                  with_pos=True),
        qg_self_attn=dict(type="QueryGroupMultiheadAttention", batch_first=True, embed_dims=e * 2, num_heads=g,
                          attn_drop=drop, dropout_layer=dict(type="Dropout", drop_prob=0.1)),
        qg_cross_attn=dict(type="QueryGroupMultiScaleDeformableAttention", batch_first=True, num_levels=lv,
                           embed_dims=e, num_points=4, residual_mode="cat"),
        refine_layer2d=dict(type="SparseBox2DRefinementModule", embed_dims=e, num_cls=ncls, with_alpha_branch=True),
        temp_graph_model=_mha(e * 2, g, drop) if temporal else None,
        graph_model=_mha(e * 2, g, drop),
        deformable_model=dict(
            type="DeformableFeatureAggregation", embed_dims=e, num_groups=g, num_levels=lv, num_cams=6,
            attn_drop=0.15, use_deformable_func=True, use_camera_embed=True, residual_mode="cat",
            kps_generator=dict(type="SparseBox3DKeyPointsGenerator", num_learnable_pts=6,
                               fix_scale=[[0, 0, 0], [0.45, 0, 0], [-0.45, 0, 0], [0, 0.45, 0], [0, -0.45, 0],
                                          [0, 0, 0.45], [0, 0, -0.45]])),
        refine_layer3d=dict(type="SparseBox3DRefinementModule", embed_dims=e, num_cls=ncls, refine_yaw=True,
                            with_quality_estimation=True),
        positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True, offset=-0.5),
        coster2d=dict(type="SparseBox2DCoster", cls_cost=dict(type="FocalLossCost", weight=2.0),
                      reg_cost=dict(type="BBoxL1Cost", weight=5.0, box_format="xywh"),
                      iou_cost=dict(type="IoUCost", iou_mode="giou", weight=2.0)),
        sampler=dict(type="SparseBox3DTargetWith2D", num_dn_groups=ndn, num_temp_dn_groups=ntdn, with_alpha_angle=True,
                     dn_noise_scale=[2.0] * 3 + [0.5] * 7, max_dn_gt=32, add_neg_dn=True, cls_weight=2.0,
                     box_weight=0.25, reg_weights=[2.0] * 3 + [0.5] * 3 + [0.0] * 4,
                     cls_wise_reg_weights={cone: [2.0, 2.0, 2.0, 1.0, 1.0, 1.0, 0.0, 0.0, 1.0, 1.0]}),
        loss_cls2d=dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=2.0),
        loss_bbox2d=dict(type="L1Loss", loss_weight=5.0),
        loss_iou2d=dict(type="GIoULoss", loss_weight=2.0),
        loss_alpha2d=dict(type="L1Loss", loss_weight=0.5),
        loss_cls=dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=2.0),
        loss_reg=dict(type="SparseBox3DLoss", loss_box=dict(type="L1Loss", loss_weight=0.25),
                      loss_centerness=dict(type="CrossEntropyLoss", use_sigmoid=True),
                      loss_yawness=dict(type="GaussianFocalLoss"), cls_allow_reverse=[CLASS_NAMES.index("barrier")]),
        decoder=dict(type="SparseBox3DDecoder"),
        reg_weights=[2.0] * 3 + [1.0] * 7,
    )
    model = dict(
        type="SimPB", use_grid_mask=True, use_deformable_func=True,
        img_backbone=dict(type="ResNet", depth=depth, num_stages=4, frozen_stages=-1, norm_eval=False, style="pytorch",
                          with_cp=True, out_indices=(0, 1, 2, 3), norm_cfg=dict(type="BN", requires_grad=True),
                          pretrained=pretrained),
        img_neck=dict(type="FPN", num_outs=lv, start_level=0, out_channels=e, add_extra_convs="on_output",
                      relu_before_extra_convs=True, in_channels=[256, 512, 1024, 2048]),
        depth_branch=dict(type="DenseDepthNet", embed_dims=e, num_depth_layers=3, loss_weight=0.2),
        head=head,
    )
    return dict(model=model, input_shape=tuple(input_shape), class_names=list(CLASS_NAMES), fp16=dict(loss_scale=32.0))
