def ensure_tuple_rep(val, dim):
    if isinstance(val, (tuple, list)):
        if len(val) != dim:
            raise ValueError("length mismatch")
        return tuple(val)
    return (val,) * dim
