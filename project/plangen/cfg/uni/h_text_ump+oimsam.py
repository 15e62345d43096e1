# Same file name as the reference experiment config (README.md:58-64); inherits the path keys.
_base_ = ["../base.py"]
out_path = "out/plangen/h_text_ump+oimsam"
