_COMP = str.maketrans("ACGTNacgtn", "TGCANtgcan")


class Seq:
    def __init__(self, data):
        self._data = str(data)

    def __str__(self):
        return self._data

    def __repr__(self):
        return f"Seq({self._data!r})"

    def __len__(self):
        return len(self._data)

    def __iter__(self):
        return iter(self._data)

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return Seq(self._data[idx])
        return self._data[idx]

    def __eq__(self, other):
        return str(self) == str(other)

    def __hash__(self):
        return hash(self._data)

    def __add__(self, other):
        return Seq(self._data + str(other))

    def __radd__(self, other):
        return Seq(str(other) + self._data)

    def find(self, sub):
        return self._data.find(str(sub))

    def count(self, sub):
        return self._data.count(str(sub))

    def reverse_complement(self):
        return Seq(self._data.translate(_COMP)[::-1])
