def natsorted(seq, key=None):
    return sorted(seq, key=key)
