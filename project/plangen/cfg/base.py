# Keys of the reference's project/plangen/cfg/base.py that the layout->image path reads
# ("Config / flags"); everything training-related is intentionally absent.
seed = 0                               # base.py:3
janus_path = "models/Janus-Pro-1B"     # base.py:8,12  (HF safetensors directory; synthetic weights if missing)
system_cls_path = "project.plangen.plangen_base"
out_path = "out/plangen"
resume = "latest"                      # base.py:47: newest checkpoint-* under out_path, a path, or None
test = False
janus_hw = 384                         # base.py:83
parallel_size = 1                      # base.py:158
cfg_weight = 5.0                       # base.py:162
temperature = 1.0
use_teacher_forcing = False            # base.py:36
use_neg_box = False                    # base.py:121
neg_prompt = ""                        # base.py:129: wrapped as wrap_uni_prompt(neg_prompt, '') for every uncond CFG row (:673)
neg_prompt_ids = None                  # optional pre-tokenised negative prompt (no tokenizer files needed)
max_new_tokens = 512                   # x2t (:513-523)
test_start = 0                         # base.py: skip batches before this index (:1135)
synthetic = False                      # True: seeded random-init weights + offline tokenizer (plumbing runs only)
max_test_len = 8                       # base.py:34: number of test batches
debug_max_seq_len = None               # base.py:135
dtype = "bf16"
test_batch_size = 8
test_data = dict(
    data_name="synthetic",             # 'synthetic' or the dataset tag used in the output path
    task_type="uni",                   # 't2i' | 'uni_2stage' | 'uni' | 'mmu' | 'plan'   (plangen_base.py:1112-1127)
    data_file=None,                    # JSONL: text rows (base_caption / gt_grounding / image_id) or pre-tokenised rows
)
