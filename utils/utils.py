"""CLI flag plumbing with the reference's parsing rules (reference utils/utils.py:74-105, :151-152)."""


def print_arguments(args):
    print("-----------  Configuration Arguments -----------")
    for arg, value in vars(args).items():
        print(f"{arg}: {value}")
    print("------------------------------------------------")


_TRUE = ("y", "yes", "t", "true", "on", "1")
_FALSE = ("n", "no", "f", "false", "off", "0")


def strtobool(val):
    """'y/yes/t/true/on/1' -> True, 'n/no/f/false/off/0' -> False (case-insensitive), anything else raises."""
    v = val.lower()
    if v in _TRUE:
        return True
    if v in _FALSE:
        return False
    raise ValueError("invalid truth value %r" % (val,))


def str_none(val):
    """The literal string 'None' means None."""
    return None if val == "None" else val


def add_arguments(argname, type, default, help, argparser, **kwargs):
    """bool flags parse through strtobool, str flags through str_none (so `--lora_model=None` works)."""
    if type == bool:
        type = strtobool
    elif type == str:
        type = str_none
    argparser.add_argument("--" + argname, default=default, type=type, help=help + " Default: %(default)s.", **kwargs)


def make_inputs_require_grad(module, input, output):
    """forward hook the reference registers on encoder.conv1 (finetune.py:173); a no-op for the HIP engine, whose
    backward is explicit, but kept so `register_forward_hook(make_inputs_require_grad)` call sites work."""
    if hasattr(output, "requires_grad_") and output.is_floating_point():
        output.requires_grad_(True)
