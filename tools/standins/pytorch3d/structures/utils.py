def list_to_padded(*a, **k):
    raise NotImplementedError("never reached on the ICP-Flow hot path")
