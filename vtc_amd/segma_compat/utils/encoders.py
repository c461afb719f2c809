"""segma.utils.encoders stand-in: label <-> logit column (reference src/pipeline/vtc.py:184,86,259-260)."""

from __future__ import annotations


class MultiLabelEncoder:
    def __init__(self, labels):
        labels = list(labels)
        if not labels or len(set(labels)) != len(labels):
            raise ValueError("labels must be a non-empty list of distinct names")
        self.labels = labels
        self._index = {l: i for i, l in enumerate(labels)}

    @property
    def n_labels(self) -> int:
        return len(self.labels)

    def __len__(self) -> int:
        return len(self.labels)

    def transform(self, label: str) -> int:
        return self._index[label]

    def inv_transform(self, index: int) -> str:
        return self.labels[int(index)]
