from .helpers import class_to_dict, export_policy_as_jit, get_load_path, get_args, set_seed, update_class_from_dict  # noqa: F401
