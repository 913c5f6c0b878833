"""soccdpt_amd — MI355X-native (gfx950) forward path of SOccDPT_V3.

Host-side mirror of the reference's Python API (`soccdpt_amd.model.SOccDPT`, `.model.loader`) over the
C-ABI library `libsoccdpt_hip.so` (include/soccdpt_hip.h).  The HIP library is the product; importing the
package does not need a GPU, running a forward does.
"""
__version__ = "0.1.0"
