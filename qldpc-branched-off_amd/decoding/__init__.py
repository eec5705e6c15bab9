"""Decoders: same call surface as the reference's ``src/decoding`` package, computed by libqldpc_hip on the GPU."""
