def clear_output(*a, **k):
    pass
