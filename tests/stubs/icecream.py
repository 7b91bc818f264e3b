"""TEST INFRASTRUCTURE: icecream.ic (debug print helper imported by the reference; not installed in this image)."""


def ic(*a, **k):
    return a[0] if len(a) == 1 else a
