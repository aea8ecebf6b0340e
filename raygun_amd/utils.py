"""Small host helpers mirrored from raygun/utils.py (reference: src/raygun/utils.py:10-18)."""


def passing_locals(local_dict):
    """Collect a constructor's arguments (and its **kwargs) into one dict — utils.py:10-18."""
    kwargs = {}
    for k, v in local_dict.items():
        if k[0] != "_" and k != "self":
            if k == "kwargs":
                kwargs.update(v)
            else:
                kwargs[k] = v
    return kwargs
