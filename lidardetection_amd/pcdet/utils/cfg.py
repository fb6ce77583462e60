"""Minimal stand-in for the EasyDict config objects the reference passes to its modules (pcdet/config.py:83-85):
attribute access + dict.get.  Only used by tests / the bench graphs; a real OpenPCDet checkout passes its own cfg."""


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v
