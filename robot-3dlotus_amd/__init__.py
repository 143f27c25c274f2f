"""lotus-hip: MI355X-native (gfx950) 3D-LOTUS policy forward/backward hot path.

Drop-in for `genrobo3d.models` of vlc-robot/robot-3dlotus on this one path: the modules in
`policy.py` / `ptv3.py` keep the reference's constructor arguments, batch dictionary, loss
dictionary and state_dict layout, and run on hand-written HIP kernels reached through the C-ABI
library `csrc/liblotus_hip.so` (declared in include/lotus_hip.h).  There is no CPU fallback:
every op raises if the library is missing.
"""
__version__ = "0.1.0"
