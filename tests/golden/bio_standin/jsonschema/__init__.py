"""No-op stand-in for `jsonschema` (not installed; no network): the reference's graph.py / fileutil.py /
appmap.py import it at module level but make_edges_from_fasta never calls it. Used only by make_golden.py."""


class ValidationError(Exception):
    pass


def validate(instance=None, schema=None, *a, **k):
    return None
