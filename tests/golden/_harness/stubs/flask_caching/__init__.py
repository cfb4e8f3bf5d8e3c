"""Harness stub for flask_caching."""


class Cache:
    def __init__(self, *a, **k):
        self._d = {}

    def init_app(self, *a, **k):
        pass

    def get(self, k):
        return self._d.get(k)

    def set(self, k, v, timeout=None):
        self._d[k] = v
