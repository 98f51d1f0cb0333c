"""Sample pairing of the gan-ves-seg task (reference data/unalignedZipDataset.py:6-59): synthetic sample i (real_A graph and
its label graph real_A_seg) with a RANDOM real scan real_B and a random background tile, drawn from Python's global
`random` stream exactly where the reference draws them."""
import random

from ..utils.enums import Phase


class UnalignedZipDataset:
    def __init__(self, data: dict, transform, phase=Phase.TRAIN) -> None:
        self.A_paths = data.get("real_A")
        self.A_seg_paths = data.get("real_A_seg")
        self.B_paths = data.get("real_B")
        self.background = data.get("background")
        self.transform = transform
        self.A_size = 0 if self.A_paths is None else len(self.A_paths)
        self.B_size = 0 if self.B_paths is None else len(self.B_paths)
        self.A_seg_size = 0 if self.A_seg_paths is None else len(self.A_seg_paths)
        self.background_size = 0 if self.background is None else len(self.background)
        self.phase = phase

    def __len__(self) -> int:
        return max(self.A_size, self.B_size)

    def __getitem__(self, index) -> dict:
        data = dict()
        if self.A_paths is not None:
            data["real_A_path"] = data["real_A"] = self.A_paths[index % self.A_size]
        if self.B_paths is not None:
            index_B = random.randint(0, self.B_size - 1) if "real_A" in data else index
            data["real_B_path"] = data["real_B"] = self.B_paths[index_B]
        if self.A_seg_paths is not None:
            data["real_A_seg_path"] = data["real_A_seg"] = self.A_seg_paths[index % self.A_size]
        if self.background is not None:
            data["background"] = self.background[random.randint(0, self.background_size - 1)]
        return self.transform(data)
