from neosr_amd.utils.misc import get_root_logger, set_random_seed, tc  # noqa: F401
