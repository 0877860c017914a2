"""Process environment the multi-GPU path needs, settable before torch (and with it the HIP runtime) is imported:
this module imports nothing but ``os`` and ``sys``."""
import os
import sys

MIN_QUEUES_MULTI_GPU = 6  # below this the step's streams share hardware queues with the communicator's


def set_hw_queues_for_multi_gpu(env=os.environ) -> None:
    """Called by the entry scripts BEFORE torch initialises the runtime, when WORLD_SIZE > 1: HIP's default of 4
    hardware queues per process is too few once a RCCL communicator exists next to the step's streams (DESIGN 6:
    1.17-1.25 ms per step instead of 0.98).

    * ``CRADLE_HW_QUEUES=<n>`` (1..32, the runtime refuses more) decides, whatever else is set;
      ``CRADLE_HW_QUEUES=keep`` leaves ``GPU_MAX_HW_QUEUES`` as the environment has it.
    * otherwise an existing ``GPU_MAX_HW_QUEUES`` of 6 or more is respected;
    * otherwise (unset, or a value the step is known to run slowly with -- some hosts export HIP's own default of 4
      for every process) 16 is used, and rank 0 says so on stderr: never a silent override."""
    if int(env.get("WORLD_SIZE", "1")) <= 1:
        return
    want, have = env.get("CRADLE_HW_QUEUES"), env.get("GPU_MAX_HW_QUEUES")
    if want == "keep":
        return
    if want is None:
        try:
            if have is not None and int(have) >= MIN_QUEUES_MULTI_GPU:
                return
        except ValueError:
            pass
        want = "16"
    try:
        n = int(want)
    except ValueError:
        raise ValueError(f"CRADLE_HW_QUEUES={want!r} is neither an integer nor 'keep'") from None
    if not 1 <= n <= 32:
        raise ValueError(f"CRADLE_HW_QUEUES={n} outside 1..32")
    if have is not None and have != str(n) and env.get("RANK", "0") == "0":
        print(f"cradle_vae_amd: GPU_MAX_HW_QUEUES={have} in the environment is too few for the multi-GPU step "
              f"(DESIGN 6); using {n}. CRADLE_HW_QUEUES=<n> chooses, CRADLE_HW_QUEUES=keep keeps the environment's value.",
              file=sys.stderr, flush=True)
    env["GPU_MAX_HW_QUEUES"] = str(n)
