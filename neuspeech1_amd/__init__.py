"""neuspeech1_amd — MI355X-native hot path of NeuSpeech (Whisper MEG->text train/decode).

csrc/      hand-written HIP kernels (gfx950) behind the C ABI in include/neuspeech_hip.h
lib.py     ctypes binding (fails loudly when the shared object is absent)
ops.py     torch-tensor front end of the ABI
engine.py  the Whisper MEG engine: forward / backward / optimizer step / decode
"""
__version__ = "0.1.0"
