"""Hyper-parameter bag of the boundary: ``create_hparams()`` returns an object whose attribute
names and values equal the reference's (/root/reference/hparams.py:1-101), because callers and
checkpoints depend on them (``hparams.batch_size``, ``hparams.max_decoder_steps`` ...).  The
``attention_location_*`` / ``attention_dim`` / ``distributed_run`` entries are dead in the
reference too and are kept only so attribute access keeps working.
"""
from types import SimpleNamespace

_EXPERIMENT = dict(epochs=500, iters_per_checkpoint=1000, seed=1234, dynamic_loss_scaling=True, fp16_run=False,
                   distributed_run=False, dist_backend="nccl", dist_url="tcp://localhost:54321",
                   cudnn_enabled=True, cudnn_benchmark=False, ignore_layers=["embedding.weight"])
_DATA = dict(load_mel_from_disk=False, training_files="filelists/ljs_audio_text_train_filelist.txt",
             validation_files="filelists/ljs_audio_text_val_filelist.txt", text_cleaners=["english_cleaners"])
_AUDIO = dict(sampling_rate=16000, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80,
              mel_fmin=0.0, mel_fmax=8000.0)
_MODEL = dict(num_init_filters=24, encoder_kernel_size=5, encoder_n_convolutions=5, encoder_embedding_dim=1024,
              n_frames_per_step=1, decoder_rnn_dim=1024, prenet_dim=256, max_decoder_steps=300, gate_threshold=0.5,
              p_attention_dropout=0.1, p_decoder_dropout=0.1, attention_rnn_dim=1024, attention_dim=128,
              attention_location_n_filters=32, attention_location_kernel_size=31, postnet_embedding_dim=512,
              postnet_kernel_size=5, postnet_n_convolutions=5)
_OPTIM = dict(use_saved_learning_rate=False, learning_rate=1e-4, weight_decay=1e-6, grad_clip_thresh=1.0,
              batch_size=64, mask_padding=True, teacher_forcing_probability=0.5)


def create_hparams():
    values = {}
    for group in (_EXPERIMENT, _DATA, _AUDIO, _MODEL, _OPTIM):
        values.update(group)
    return SimpleNamespace(**values)
