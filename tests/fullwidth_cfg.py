"""Janus-Pro-1B WIDTH on 2 layers: the model config of the full-width fixtures (oracle/make_golden.py::FULLW), shared by the CPU
and GPU tests (no GPU import here)."""
FULLW = dict(hidden=2048, inter=5632, n_layers=2, n_heads=16, head_dim=128, vocab=4096,
             img_vocab=16384, img_dim=8, grid=24, gen_head_dim=2048, vq_ch=64,
             vq_ch_mult=(1, 2, 2), vq_z=64, eos_id=7, pad_id=3,
             vit_width=128, vit_layers=2, vit_heads=2, vit_mlp=256, vit_patch=8, vit_img=64)
