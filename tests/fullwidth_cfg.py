"""Janus-Pro-1B WIDTH on 2 layers: the model config of the full-width fixtures (oracle/make_golden.py::FULLW), shared by the CPU
and GPU tests (no GPU import here)."""
FULLW = dict(hidden=2048, inter=5632, n_layers=2, n_heads=16, head_dim=128, vocab=4096,
             img_vocab=16384, img_dim=8, grid=24, gen_head_dim=2048, vq_ch=64,
             vq_ch_mult=(1, 2, 2), vq_z=64, eos_id=7, pad_id=3,
             vit_width=128, vit_layers=2, vit_heads=2, vit_mlp=256, vit_patch=8, vit_img=64)

"""SigLIP-L at its PRODUCTION shape (siglip_vit.py:628-637: width 1024, 16 heads x 64, MLP 4096, patch 16, 384^2 -> 576 tokens) on 2
layers, in front of a one-layer language model of Janus width (aligner 1024 -> 2048 -> 2048): oracle/make_golden.py::VISW."""
VISW = dict(hidden=2048, inter=512, n_layers=1, n_heads=16, head_dim=128, vocab=512,
            img_vocab=256, img_dim=8, grid=8, gen_head_dim=256, vq_ch=64,
            vq_ch_mult=(1, 2, 2), vq_z=64, eos_id=7, pad_id=3,
            vit_width=1024, vit_layers=2, vit_heads=16, vit_mlp=4096, vit_patch=16, vit_img=384)


def siglip_fullwidth_images(n=2, seed=41, S=384):
    """The fixture's seeded images (same arithmetic as oracle/make_golden.py::siglip_fullwidth_images; 3.5 MB not committed,
    the fixture stores their |sum| so drift is caught)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, S), torch.linspace(0, 1, S), indexing="ij")
    img = torch.rand(n, 3, S, S, generator=g) * 2 - 1
    for i in range(n):
        for c in range(3):
            f = torch.rand(4, generator=g) * 6 + 1
            img[i, c] = 0.5 * img[i, c] + 0.5 * torch.sin(f[0] * yy * 3.1 + f[1]) * torch.cos(f[2] * xx * 2.7 + f[3])
    return img.clamp(-1, 1)

"""Janus-Pro-1B width on 2 layers with the REAL vocabulary (102 400 rows, EOS 100 001): oracle/make_golden.py::FULLV."""
FULLV = dict(FULLW, vocab=102400, eos_id=100001, pad_id=100002)
