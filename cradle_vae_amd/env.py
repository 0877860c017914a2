"""Process environment the multi-GPU path needs, settable before torch (and with it the HIP runtime) is imported:
this module imports nothing but ``os``."""
import os


def set_hw_queues_for_multi_gpu(env=os.environ) -> None:
    """Called by the entry scripts BEFORE torch initialises the runtime, when WORLD_SIZE > 1: HIP's default of 4
    hardware queues per process is too few once a RCCL communicator exists next to the step's streams.
    An existing GPU_MAX_HW_QUEUES is respected; CRADLE_HW_QUEUES overrides the default of 16 and must lie in 1..32
    (the runtime refuses more)."""
    if int(env.get("WORLD_SIZE", "1")) <= 1 or "GPU_MAX_HW_QUEUES" in env:
        return
    want = env.get("CRADLE_HW_QUEUES", "16")
    try:
        n = int(want)
    except ValueError:
        raise ValueError(f"CRADLE_HW_QUEUES={want!r} is not an integer") from None
    if not 1 <= n <= 32:
        raise ValueError(f"CRADLE_HW_QUEUES={n} outside 1..32")
    env["GPU_MAX_HW_QUEUES"] = str(n)
