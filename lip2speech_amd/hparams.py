"""Hyper-parameter bag of the boundary: ``create_hparams()`` returns an object whose attribute names and values equal the
reference's (/root/reference/hparams.py:1-101), because callers and checkpoints depend on them (``hparams.batch_size``,
``hparams.max_decoder_steps`` ...).  Kept as one alphabetical ``name = literal`` table; the ``attention_*`` / ``distributed_run`` /
``dist_*`` / ``text_cleaners`` / file-list entries are dead in the reference too and exist only so that attribute
access keeps working.
"""
import ast
from types import SimpleNamespace

_TABLE = """
    attention_dim = 128
    attention_location_kernel_size = 31
    attention_location_n_filters = 32
    attention_rnn_dim = 1024
    batch_size = 64
    cudnn_benchmark = False
    cudnn_enabled = True
    decoder_rnn_dim = 1024
    dist_backend = 'nccl'
    dist_url = 'tcp://localhost:54321'
    distributed_run = False
    dynamic_loss_scaling = True
    encoder_embedding_dim = 1024
    encoder_kernel_size = 5
    encoder_n_convolutions = 5
    epochs = 500
    filter_length = 1024
    fp16_run = False
    gate_threshold = 0.5
    grad_clip_thresh = 1.0
    hop_length = 256
    ignore_layers = ['embedding.weight']
    iters_per_checkpoint = 1000
    learning_rate = 0.0001
    load_mel_from_disk = False
    mask_padding = True
    max_decoder_steps = 300
    mel_fmax = 8000.0
    mel_fmin = 0.0
    n_frames_per_step = 1
    n_mel_channels = 80
    num_init_filters = 24
    p_attention_dropout = 0.1
    p_decoder_dropout = 0.1
    postnet_embedding_dim = 512
    postnet_kernel_size = 5
    postnet_n_convolutions = 5
    prenet_dim = 256
    sampling_rate = 16000
    seed = 1234
    teacher_forcing_probability = 0.5
    text_cleaners = ['english_cleaners']
    training_files = 'filelists/ljs_audio_text_train_filelist.txt'
    use_saved_learning_rate = False
    validation_files = 'filelists/ljs_audio_text_val_filelist.txt'
    weight_decay = 1e-06
    win_length = 1024
"""


def create_hparams():
    pairs = (line.split("=", 1) for line in _TABLE.strip().splitlines())
    return SimpleNamespace(**{name.strip(): ast.literal_eval(value.strip()) for name, value in pairs})
