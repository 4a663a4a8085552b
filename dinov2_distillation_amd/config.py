"""Tiny attribute/dict config (the reference uses OmegaConf, absent here) + the derivations of train.py:86-122,188-190."""


class Cfg(dict):
    """dict with attribute access, like an OmegaConf node for the keys this path reads"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_cfg(o):
    if isinstance(o, dict):
        return Cfg({k: to_cfg(v) for k, v in o.items()})
    if isinstance(o, (list, tuple)):
        return [to_cfg(v) for v in o]
    return o


TEACHER_DIMS = {"dinov2_vits14": 384, "dinov2_vitb14": 768, "dinov2_vitl14": 1024, "dinov2_vitg14": 1536}


def make_config(teacher="dinov2_vits14", student="convnext_tiny", crop=224, student_keys=("res5", "res4"),
                heads=(16, 24), lr=1e-3):
    """config/config.yaml defaults for the chosen pair, with ``_handle_config`` (train.py:86-122) applied.  Synthetic runs have
    no checkpoint files (the reference's are git-ignored): ``allow_missing_checkpoint`` is set, a real config leaves it out."""
    n = crop // 14
    d = TEACHER_DIMS[teacher]

    def spec(name, h, selfq):
        return dict(type="scalekd", weight=1.0, kwargs=dict(alpha=[0.08, 0.06], window_shapes=[1, 1], self_query=selfq,
                                                           softmax_scale=[5.0, 5.0], num_heads=h, name=name,
                                                           teacher_dims=d, pos_dims=d, pos_hw=[n, n], query_hw=[n, n]))

    return to_cfg(dict(
        student=dict(model_name=student, student_keys=list(student_keys), checkpoint_path=None, allow_missing_checkpoint=True),
        teacher=dict(model_name=teacher, out_dim=d, teacher_key="feature_map", n_patches=[n, n]),
        data_transform=dict(global_crops_size=[crop, crop]),
        optimizer=dict(type="AdamW", kwargs=dict(lr=lr, betas=[0.9, 0.999], weight_decay=0.01)),
        loss=dict(losses=[spec("scalekd_res4", heads[0], True), spec("scalekd_res5", heads[1], False)]),
        train=dict(gradient_clip_val=1.0),
    ))


def set_student_dims(cfg, student):
    """train.py:188-190"""
    for loss in cfg.loss.losses:
        if loss.type == "scalekd":
            loss.kwargs.student_dims = int(student.feature_channels[loss.kwargs.name.split("_")[1]])
    return cfg
