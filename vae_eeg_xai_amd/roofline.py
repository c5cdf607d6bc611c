"""Live roofline measurement of the dominant kernel (filled in after profiling)."""


def dominant_kernel_roofline(net, batch):
    return None
