"""Oracle shim (TEST INFRASTRUCTURE ONLY): minimal stand-in for the un-installed
`addict` package.  The reference only uses `addict.Dict(**kwargs)` as an attribute-access
dict (models/modeling_distributed_gpt3.py:1618)."""


class Dict(dict):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v
