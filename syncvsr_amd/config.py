"""Configuration objects for the SyncVSR hot path.

The reference reads an OmegaConf tree (``OmegaConf.merge(OmegaConf.load(argv[1]), OmegaConf.from_cli())``,
reference ``LRW/video/src/train.py:51``) with attribute access (``config.model.bert.dim``) and splats
``config.model.bert`` into ``BertConfig(**...)`` (``LRW/video/src/lightning.py:92``).  omegaconf is not
available on the target image, so this is a tiny attribute-dict with the same access pattern, a PyYAML
loader and ``a.b=c`` dot-list overrides.
"""
from __future__ import annotations

import copy
from typing import Any, Iterable


class Config(dict):
    """dict with attribute access, nested; usable with ``**`` splatting like a DictConfig."""

    def __init__(self, *args: Any, **kwargs: Any):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, k: str, v: Any) -> None:
        if isinstance(v, dict) and not isinstance(v, Config):
            v = Config(v)
        super().__setitem__(k, v)

    def __getattr__(self, k: str) -> Any:
        try:
            return self[k]
        except KeyError as e:  # mirror OmegaConf's error type loosely
            raise AttributeError(k) from e

    def __setattr__(self, k: str, v: Any) -> None:
        self[k] = v

    def get_path(self, path: str, default: Any = None) -> Any:
        node: Any = self
        for part in path.split("."):
            if not isinstance(node, dict) or part not in node:
                return default
            node = node[part]
        return node

    def set_path(self, path: str, value: Any) -> None:
        parts = path.split(".")
        node = self
        for part in parts[:-1]:
            if part not in node or not isinstance(node[part], dict):
                node[part] = Config()
            node = node[part]
        node[parts[-1]] = value

    def copy(self) -> "Config":  # deep
        return Config(copy.deepcopy(dict(self)))


def _parse_scalar(text: str) -> Any:
    import yaml

    return yaml.safe_load(text)


def apply_dotlist(cfg: Config, overrides: Iterable[str]) -> Config:
    """``a.b=c`` overrides, the OmegaConf.from_cli() subset the reference uses (train.py:51)."""
    for item in overrides:
        if "=" not in item:
            raise ValueError(f"override {item!r} is not of the form key=value")
        key, val = item.split("=", 1)
        cfg.set_path(key, _parse_scalar(val))
    return cfg


def load_yaml(path: str, overrides: Iterable[str] = ()) -> Config:
    import yaml

    with open(path) as f:
        cfg = Config(yaml.safe_load(f))
    return apply_dotlist(cfg, overrides)


def default_lrw_config(**kw: Any) -> Config:
    """BASELINE.json config 2: ResNet18 + 6-layer 512-d BERT-style encoder + vq audio head.

    Keys follow the reference yaml (``LRW/video/config/bert-12l-512d_LRW_96_bf16_rrc_noWB.yaml``)
    with the ``type: huggingface`` encoder branch (``lightning.py:90-93``), whose BertConfig keys
    are carried in the same ``model.bert`` sub-tree.
    """
    cfg = Config(
        data=dict(use_word_boundary=False, input_size=88),
        model=dict(
            name="transformer",
            resnet="resnet18",
            wav2vec=dict(path="./vq-wav2vec_kmeans.pt"),
            bert=dict(
                type="huggingface",
                dim=512,
                hidden_size=512,
                num_hidden_layers=6,
                num_attention_heads=8,
                intermediate_size=2048,
                hidden_dropout_prob=0.0,
                attention_probs_dropout_prob=0.0,
                layer_norm_eps=1e-12,
                max_position_embeddings=512,
                type_vocab_size=2,
                emb_dropout=0.0,
                num_labels=500,
            ),
        ),
        optim=dict(
            optimizer=dict(lr=1e-4, betas=[0.9, 0.999], eps=1e-6, weight_decay=0.01),
            scheduler=dict(name="cosine", num_warmup_steps=15000, num_training_steps=270000),
            lambda_audio=10.0,
        ),
        train=dict(
            batch_size=32,
            gradient_clip_val=1.0,
            label_smoothing=0.0,
            use_cutmix=False,
            precision="bf16",
        ),
    )
    for k, v in kw.items():
        cfg.set_path(k.replace("__", "."), v)
    return cfg


def xtransformers_lrw_config(word_boundary: bool = True, **kw: Any) -> Config:
    """The encoder the reference's shipped LRW yamls select (``LRW/video/config/bert-12l-512d_LRW_96_bf16_rrc_{WB,noWB}.yaml:17-30``):
    ``type: x-transformers``, depth 12, RMSNorm, GEGLU feed-forward, rotary embedding, layer-drop 0.2, ff-dropout 0.3; with
    ``use_word_boundary`` the encoder is 513 wide (lightning.py:49,145)."""
    cfg = default_lrw_config()
    cfg.data["use_word_boundary"] = bool(word_boundary)
    cfg.model["bert"] = Config(
        type="x-transformers", num_tokens=1, dim=512, depth=12, heads=8, emb_dropout=0.0, attn_dropout=0.0, layer_dropout=0.2,
        ff_dropout=0.3, use_rmsnorm=True, ff_glu=True, rotary_pos_emb=True, num_labels=500)
    for k, v in kw.items():
        cfg.set_path(k.replace("__", "."), v)
    return cfg


def audio_codec_dims(path: str) -> tuple[str, int, int, int]:
    """(codec, audio_alignment A, vq_groups G, audio_vocab_size V) — reference lightning.py:58-67."""
    if "vq" in path:
        return "vq", 4, 2, 320
    if "wav2vec2" in path:
        return "wav2vec2", 2, 2, 640
    raise ValueError(f"cannot infer audio codec from wav2vec path {path!r}")
