#!/usr/bin/env python
"""LSTM-cell launch duration (chain of 600 launches between one HIP-event pair) per block shape and batch rows."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
sd = synth.synth_state_dict()
nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
for B in (32, 64, 128, 256, 512):
    nm.workspace(B, 29, 96, 96, 300, torch.device("cuda", 0))
    row = []
    for shape in (11, 21, 22, 42):
        for jb in (4, 2):
            nm.set_option("skinny_rc", shape); nm.set_option("skinny_rc_jb", jb)
            us = min(nm.lstm_cell_chain_us(B, 300) for _ in range(3))
            fl = 2 * B * 2048 * (1536 + 1024) / 2
            row.append(f"{shape}/jb{jb}: {us:6.2f} us ({fl/us/1e6:5.1f} TF)")
    print(f"B={B:3d}  " + "  ".join(row), flush=True)
