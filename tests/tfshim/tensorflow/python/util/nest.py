"""TEST INFRASTRUCTURE: tensorflow.python.util.nest.map_structure over tuples / namedtuples / lists / dicts."""


def _is_seq(x):
    return isinstance(x, (tuple, list, dict))


def map_structure(func, *structure, **kw):
    s0 = structure[0]
    if not _is_seq(s0):
        return func(*structure)
    if isinstance(s0, dict):
        return type(s0)((k, map_structure(func, *[s[k] for s in structure])) for k in s0)
    for s in structure[1:]:
        if len(s) != len(s0):
            raise ValueError("The two structures don't have the same nested structure")
    items = [map_structure(func, *[s[i] for s in structure]) for i in range(len(s0))]
    if hasattr(s0, "_fields"):              # namedtuple
        return type(s0)(*items)
    return type(s0)(items)


def flatten(structure):
    if not _is_seq(structure):
        return [structure]
    vals = structure.values() if isinstance(structure, dict) else structure
    out = []
    for v in vals:
        out += flatten(v)
    return out
