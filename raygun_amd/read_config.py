"""JSON-with-comments config loader — same semantics as raygun.read_config (reference: src/raygun/read_config.py:30-95):
  * a dict is returned as is;
  * `//` comments are stripped (the reference uses jsmin);
  * "include_config" chains are followed, later files are overridden by earlier ones (".." expands to the
    grand-parent directory of the including file);
  * string values of the form "#expr#" are eval'd with torch / np / numpy / partial in scope, and "...#expr#..."
    substrings are spliced; "$working_dir" expands to the directory of the top-level config file.
"""
import json
import os
import re
from functools import partial  # noqa: F401  (available to #...# expressions)

import numpy  # noqa: F401
import numpy as np  # noqa: F401

try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

_COMMENT = re.compile(r'("(?:\\.|[^"\\])*")|//[^\n]*|/\*.*?\*/', re.S)


def strip_comments(text):
    return _COMMENT.sub(lambda m: m.group(1) or "", text)


def load_json_file(path):
    with open(path) as f:
        return json.loads(strip_comments(f.read()))


def _splice(arg):
    parts = []
    while arg.count("#") >= 2:
        a = arg.find("#")
        b = arg.find("#", a + 1)
        parts.append(arg[:a])
        parts.append(str(eval(arg[a + 1:b])))
        arg = arg[b + 1:]
    return "".join(parts) + arg


def eval_args(config, file):
    for k, v in config.items():
        if isinstance(v, dict):
            config[k] = eval_args(v, file)
        elif isinstance(v, str) and len(v) > 0:
            if "$working_dir" in v:
                v = v.replace("$working_dir", os.path.dirname(file))
            if v[0] == "#" and v[-1] == "#" and len(v) > 1:
                v = eval(v[1:-1])
            elif v.count("#") > 0 and v.count("#") % 2 == 0:
                v = _splice(v)
            config[k] = v
    return config


def read_config(file):
    if isinstance(file, dict):
        return file
    configs = [load_json_file(file)]
    last = file
    while "include_config" in configs[-1]:
        inc = configs[-1]["include_config"]
        if ".." in inc:
            inc = inc.replace("..", os.path.dirname(os.path.dirname(last)))
        configs.append(load_json_file(inc))
        last = inc
    config = {}
    for c in configs[::-1]:
        config.update(**c)
    return eval_args(config, file)
