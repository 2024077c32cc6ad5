"""YAML option loading — the reference uses OmegaConf.to_container(OmegaConf.load(path), resolve=True)
(train_edlora.py:31); omegaconf is not installed, so PyYAML + a minimal `${a.b}` interpolation resolver (the
shipped option files use no interpolation). The full option surface of options/train/**.yml is accepted; keys that
steer out-of-scope subsystems (image transforms, visual validation) are parsed and kept."""
import re

import yaml


def _resolve(node, root):
    if isinstance(node, dict):
        return {k: _resolve(v, root) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if isinstance(node, str):
        def sub(m):
            cur = root
            for part in m.group(1).split('.'):
                cur = cur[part]
            return str(cur)
        return re.sub(r'\$\{([^}]+)\}', sub, node)
    return node


def load_options(path):
    with open(path, 'r') as f:
        opt = yaml.safe_load(f)
    return _resolve(opt, opt)


def dict2str(opt, indent=1):
    msg = '\n'
    for k, v in opt.items():
        if isinstance(v, dict):
            msg += ' ' * (indent * 2) + k + ':[' + dict2str(v, indent + 1) + ' ' * (indent * 2) + ']\n'
        else:
            msg += ' ' * (indent * 2) + k + ': ' + str(v) + '\n'
    return msg
