"""Shared test helpers: small configurations and deterministic, non-trivial weights."""
import torch

from fiery_amd.config import get_preset_cfg


def tiny_cfg(preset='baseline.yml', bev=16, **overrides):
    """A configuration with the reference's structure but a small image/BEV so the CPU tiers are fast."""
    half = bev / 2.0
    opts = ['IMAGE.FINAL_DIM', '(64, 96)', 'LIFT.X_BOUND', f'[-{half}, {half}, 1.0]', 'LIFT.Y_BOUND', f'[-{half}, {half}, 1.0]',
            'LIFT.D_BOUND', '[2.0, 10.0, 2.0]']
    for k, v in overrides.items():
        opts += [k, str(v)]
    return get_preset_cfg(preset, opts)


from fiery_amd.synthetic import randomise_weights  # noqa: E402,F401  (lives with the other synthetic-input generators)


def forward_case(cfg, rf, n_future, depth, bev_size, B, n_cam, with_labels=False, with_noise=False, seed=0):
    """Seeded inputs of one hot-path forward: lifted features (B, rf, n, C, D, fh, fw) + cameras + ego-motion."""
    from fiery_amd.synthetic import make_inputs, make_lifted_features
    fh, fw = cfg.IMAGE.FINAL_DIM[0] // cfg.MODEL.ENCODER.DOWNSAMPLE, cfg.IMAGE.FINAL_DIM[1] // cfg.MODEL.ENCODER.DOWNSAMPLE
    C = cfg.MODEL.ENCODER.OUT_CHANNELS
    _, K, E, ego = make_inputs(B, rf + n_future, n_cam, with_image=False, seed=seed)
    _, _, lifted = make_lifted_features(B * rf * n_cam, C, depth, (fh, fw), seed=seed + 1)
    lifted = lifted.view(B, rf, n_cam, C, depth, fh, fw)
    labels = noise = None
    if with_labels:
        labels = torch.randn(B, 1 + n_future, 6, *bev_size, generator=torch.Generator().manual_seed(seed + 5))
    if with_noise:
        noise = torch.randn(B, 1, cfg.MODEL.DISTRIBUTION.LATENT_DIM, generator=torch.Generator().manual_seed(seed + 6))
    return lifted, K, E, ego, labels, noise
