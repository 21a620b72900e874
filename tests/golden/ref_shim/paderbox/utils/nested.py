"""Container recursion helpers (stand-ins for paderbox.utils.nested)."""
import dataclasses


def flatten(d, sep='.', flat_type=dict):
    out = {}

    def walk(prefix, value):
        if isinstance(value, flat_type) and len(value):
            for k, v in value.items():
                walk(prefix + (k,), v)
        else:
            out[prefix if sep is None else sep.join(str(p) for p in prefix)] = value

    for key, value in d.items():
        walk((key,), value)
    return out


def deflatten(d, sep='.', maxdepth=-1):
    out = {}
    for key, value in d.items():
        if sep is None:
            parts = list(key)
        elif maxdepth >= 0:
            parts = key.split(sep, maxdepth)
        else:
            parts = key.split(sep)
        node = out
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = value
    return out


def nested_op(func, arg1, *args, broadcast=False, handle_dataclass=False, keep_type=True,
              mapping_type=dict, sequence_type=(tuple, list)):
    kw = dict(handle_dataclass=handle_dataclass)
    if isinstance(arg1, mapping_type):
        return arg1.__class__(
            {k: nested_op(func, arg1[k], *[a[k] for a in args], **kw) for k in arg1})
    if isinstance(arg1, sequence_type):
        return arg1.__class__([nested_op(func, *a, **kw) for a in zip(arg1, *args)])
    if handle_dataclass and dataclasses.is_dataclass(arg1) and not isinstance(arg1, type):
        return arg1.__class__(**{
            f.name: nested_op(func, getattr(arg1, f.name),
                              *[getattr(a, f.name) for a in args], **kw)
            for f in dataclasses.fields(arg1)})
    return func(arg1, *args)


def nested_merge(default, *updates, **kwargs):
    out = dict(default)
    for u in updates:
        for k, v in u.items():
            if isinstance(v, dict) and isinstance(out.get(k), dict):
                out[k] = nested_merge(out[k], v)
            else:
                out[k] = v
    return out
