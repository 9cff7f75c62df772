"""Nested-structure helpers with tf.nest ordering rules.

The reference flattens namedtuples in field order and dicts in sorted-key order
(tf.nest; tf_agents/replay_buffers/table.py:54-77 allocates one variable per flattened leaf), so
the canonical leaf order of a Trajectory is step_type, observation, action, policy_info leaves
(sorted keys), next_step_type, reward, discount.  Mirrors the subset of
tf_agents/utils/nest_utils.py the hot path needs.
"""
import collections.abc

import torch


def _is_namedtuple(x):
    return isinstance(x, tuple) and hasattr(x, "_fields")


def is_nested(x):
    return isinstance(x, (tuple, list, dict))


def _children(x):
    if isinstance(x, dict):
        return [x[k] for k in sorted(x)]
    return list(x)


def flatten(structure):
    """Leaves of `structure` in tf.nest order.  Empty tuples/dicts contribute nothing."""
    if not is_nested(structure):
        return [structure]
    out = []
    for c in _children(structure):
        out.extend(flatten(c))
    return out


def _rebuild(structure, children):
    if isinstance(structure, dict):
        return type(structure)((k, c) for k, c in zip(sorted(structure), children))
    if _is_namedtuple(structure):
        return type(structure)(*children)
    return type(structure)(children)


def _pack(s, flat, pos):
    if not is_nested(s):
        v = flat[pos[0]]
        pos[0] += 1
        return v
    return _rebuild(s, [_pack(c, flat, pos) for c in _children(s)])


def pack_sequence_as(structure, flat):
    # (module-level recursion: a local `def rec` that calls itself is a reference cycle through
    # its own closure cell, and these cycles held the device tensors of every packed nest until
    # the cyclic collector ran -- tools/lifetime_probe.py)
    flat = list(flat)
    pos = [0]
    out = _pack(structure, flat, pos)
    if pos[0] != len(flat):
        raise ValueError(f"pack_sequence_as: structure has {pos[0]} leaves, got {len(flat)}")
    return out


def map_structure(fn, *structures):
    first = structures[0]
    for s in structures[1:]:
        assert_same_structure(first, s)
    if not is_nested(first):
        return fn(*structures)
    kids = [_children(s) for s in structures]
    return _rebuild(first, [map_structure(fn, *cs) for cs in zip(*kids)])


def _mismatch(x, y):
    """None if the nests agree in structure, else the reason of the first difference."""
    if is_nested(x) != is_nested(y):
        return "one is a leaf, the other a sequence"
    if not is_nested(x):
        return None
    if isinstance(x, dict) != isinstance(y, dict):
        return "dict vs non-dict"
    if isinstance(x, dict):
        if sorted(x) != sorted(y):
            return f"dict keys differ {sorted(x)} vs {sorted(y)}"
    else:
        if _is_namedtuple(x) != _is_namedtuple(y):
            return "namedtuple vs plain sequence"
        if _is_namedtuple(x) and type(x).__name__ != type(y).__name__:
            return f"namedtuple types differ {type(x).__name__} vs {type(y).__name__}"
        if len(x) != len(y):
            return f"lengths differ {len(x)} vs {len(y)}"
    for cx, cy in zip(_children(x), _children(y)):
        why = _mismatch(cx, cy)
        if why is not None:
            return why
    return None


def assert_same_structure(a, b, message=None):
    """Raises ValueError if the two nests differ in structure (types of sequences, dict keys)."""
    why = _mismatch(a, b)
    if why is not None:
        raise ValueError((message + ": " if message else "") +
                         f"The two structures do not match: {why}.\nFirst: {_brief(a)}\n"
                         f"Second: {_brief(b)}")


def _brief(x):
    return map_structure(lambda l: getattr(l, "shape", l) if not isinstance(l, type) else l, x) \
        if is_nested(x) else repr(x)


def has_lists(structure):
    if isinstance(structure, list):
        return True
    if is_nested(structure):
        return any(has_lists(c) for c in _children(structure))
    return False


def get_outer_shape(nested_tensor, spec):
    """Leading dims of the first leaf beyond its spec's shape."""
    t = flatten(nested_tensor)
    s = flatten(spec)
    if not t:
        return ()
    rank = len(s[0].shape)
    shp = tuple(t[0].shape)
    return shp[:len(shp) - rank]


def get_outer_rank(nested_tensor, spec):
    """Number of outer (batch/time) dims; validates that all leaves agree."""
    assert_same_structure(nested_tensor, spec)
    ranks = set()
    for t, s in zip(flatten(nested_tensor), flatten(spec)):
        r = t.dim() - len(s.shape)
        if r < 0 or tuple(t.shape[r:]) != tuple(s.shape):
            raise ValueError(f"tensor shape {tuple(t.shape)} is not compatible with spec shape "
                             f"{tuple(s.shape)}")
        ranks.add(r)
    if len(ranks) > 1:
        raise ValueError(f"leaves disagree on outer rank: {sorted(ranks)}")
    return ranks.pop() if ranks else 0


def is_batched_nested_tensors(tensors, specs, num_outer_dims=1):
    r = get_outer_rank(tensors, specs)
    if r == num_outer_dims:
        return True
    if r == num_outer_dims - 1:
        return False
    raise ValueError(f"Received tensors with outer rank {r}, expected {num_outer_dims} or "
                     f"{num_outer_dims - 1}")


def batch_nested_tensors(tensors, specs=None):
    return map_structure(lambda t: t.unsqueeze(0), tensors)


def unbatch_nested_tensors(tensors, specs=None):
    return map_structure(lambda t: t.squeeze(0), tensors)


def to_tensor(x, dtype=None, device=None):
    if isinstance(x, torch.Tensor):
        return x
    return torch.as_tensor(x, dtype=dtype, device=device)
