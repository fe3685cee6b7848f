class BgzfWriter:  # names only; never instantiated by make_golden.py
    pass


class BgzfReader:
    pass


class BgzfBlocks:
    pass
