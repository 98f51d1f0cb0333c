"""Dotted-key CLI overrides of nested YAML configs (behaviour of the reference's utils/config_overrides.py:18-62):
`--A.B.c value`, `--A.B.c=value`, bare `--A.B.c` (-> true); values are YAML-parsed; keys without a dot are left
to argparse."""
import yaml


def parse_cli_overrides(unknown_args):
    out, i = [], 0
    while i < len(unknown_args):
        tok = unknown_args[i]
        if not isinstance(tok, str) or not tok.startswith("--"):
            i += 1
            continue
        body = tok[2:]
        if "=" in body:
            out.append(tuple(body.split("=", 1)))
            i += 1
        elif i + 1 < len(unknown_args) and isinstance(unknown_args[i + 1], str) and not unknown_args[i + 1].startswith("--"):
            out.append((body, unknown_args[i + 1]))
            i += 2
        else:
            out.append((body, "true"))
            i += 1
    return out


def apply_cli_overrides_from_unknown_args(config, unknown_args):
    for key, raw in parse_cli_overrides(unknown_args):
        if "." not in key:
            continue
        node = config
        parts = key.split(".")
        for p in parts[:-1]:
            if p not in node or not isinstance(node[p], dict):
                node[p] = {}
            node = node[p]
        try:
            node[parts[-1]] = yaml.safe_load(raw)
        except Exception:
            node[parts[-1]] = raw
