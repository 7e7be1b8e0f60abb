"""Experiment helper of tools/pipe_race.py and tools/daf_stress.py."""
import torch


def cu_masked_streams(device, every):
    """(backbone stream, decoder stream) on DISJOINT sets of compute units: the decoder stream gets every `every`-th CU
    of the mask (256 / every of the 256 CUs), the backbone stream all the others (hipExtStreamCreateWithCUMask; torch
    sees them as external streams). Waves of the two hardware queues then never share a CU."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    n_cu = torch.cuda.get_device_properties(device).multi_processor_count
    words = (n_cu + 31) // 32
    head_bits = [i for i in range(n_cu) if i % every == 0]
    masks = []
    for bits in ([i for i in range(n_cu) if i % every != 0], head_bits):
        m = (ctypes.c_uint32 * words)()
        for i in bits:
            m[i // 32] |= 1 << (i % 32)
        masks.append(m)
    out = []
    with torch.cuda.device(device):
        for m in masks:
            h = ctypes.c_void_p()
            err = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(words), m)
            if err != 0:
                raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({err})")
            out.append(torch.cuda.ExternalStream(h.value, device=device))
    return out[0], out[1]
