"""Drop-in name for the reference's ``train_utils`` package: only ``train_utils.losses`` is on the hot path (tensorboard logging and
plotting are out of scope)."""
