from .Seq import Seq


class SeqRecord:
    def __init__(self, seq, id="<unknown id>", name="", description=""):
        self.seq = seq if isinstance(seq, Seq) else Seq(seq)
        self.id = id
        self.name = name
        self.description = description

    def __len__(self):
        return len(self.seq)
