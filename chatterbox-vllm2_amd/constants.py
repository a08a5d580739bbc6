"""Constants of the T3 path.  Reference: src/chatterbox_vllm/models/t3/t3.py:38-49,
models/t3/modules/t3_config.py:1-38, t3-model/config.json:1-33 (paths relative to the reference repo)."""

PREFILL_COND_START_TOKEN = 695   # t3.py:38
PREFILL_COND_END_TOKEN = 696     # t3.py:39
PREFILL_END_TOKEN = 697          # t3.py:40
CONDITIONING_SIZE = 34           # t3.py:42
SPEECH_TOKEN_OFFSET = 2500       # t3.py:49

START_SPEECH_TOKEN = 6561        # t3_config.py:8
STOP_SPEECH_TOKEN = 6562         # t3_config.py:9
SPEECH_VOCAB = 8194              # t3_config.py:10
MAX_TEXT_POS = 2050              # t3.py:280  (max_text_tokens + 2)
MAX_SPEECH_POS = 4100            # t3.py:283  (max_speech_tokens + 2 + 2)
TEXT_VOCAB_EN = 704              # t3.py:270
TEXT_VOCAB_MTL = 2454

HIDDEN = 1024                    # t3.py:263 (config.json says 2048; reset to 1024)
N_HEADS = 16                     # config.json:16
HEAD_DIM = 64                    # config.json:8
FFN = 4096                       # config.json:12
N_LAYERS = 30                    # config.json:17
RMS_EPS = 1e-5                   # config.json:20
S3_TOKEN_RATE = 25               # models/s3tokenizer/s3tokenizer.py:18 (speech tokens per second)

ATTN_CHUNK_TOKENS = 64           # attention chunk (numerics contract)
KV_BLOCK_TOKENS = 256            # physical KV block: 4 chunks, 32 KiB K + 32 KiB V contiguous per head
KV_BYTES_PER_TOKEN_PER_STREAM = 2 * N_LAYERS * N_HEADS * HEAD_DIM * 2   # 122 880
