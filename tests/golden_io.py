"""Loader for tests/golden/*.npz (written by tools/make_golden.py from the real reference)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

ERASE_CASES = ["erase_n2p3_d768", "erase_n50_d768", "erase_n1000p500_d768", "erase_quirks_d768",
               "erase_n12p4_d1024", "erase_n36p4_d2048", "erase_n300p100_d768"]
DEBIAS_CASES = ["debias_n4x2_d768", "debias_n36x2_d2048"]
# the reference's string-keyed, in-place-drifted cache (uce_sd_debias.py:69-88, 122-127): duplicate edit concept, a string in
# both the edit and the preserve list, debias concepts that are edit concepts
DEBIAS_ALIAS_CASES = ["debias_alias_dupedit_d768", "debias_alias_editpres_d768", "debias_alias_editisdebias_d768"]
CLI_CASES = ["cli_erase_art_expand", "cli_erase_object_default", "cli_erase_object_expand_guided"]
SDPA_CASES = ["sdpa_Lq4096_dh40", "sdpa_Lq1024_dh80", "sdpa_Lq256_dh160", "sdpa_Lq64_dh160"]


class Case:
    def __init__(self, name):
        self.name = name
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.z = z
        self.meta = json.loads(str(z["meta"]))
        self.n_modules = len([k for k in z.files if k.startswith("W_old_")])

    def arr(self, key):
        return self.z[key]

    def t(self, key):
        return torch.from_numpy(self.z[key])

    def w_old(self):
        return [self.t(f"W_old_{i}") for i in range(self.n_modules)]

    def w_ref32(self):
        return [self.t(f"W_ref32_{i}") for i in range(self.n_modules)]

    def w_exact64(self):
        return [self.t(f"W_exact64_{i}") for i in range(self.n_modules)]


def rows(a):
    """[N,d] array -> list of [1,d] tensors (what the reference's embedding dict holds)."""
    return [torch.from_numpy(np.ascontiguousarray(r[None])) for r in a]


def keyed_embeds(c: "Case"):
    """{string: [1, d] tensor} of a debias case (the reference's `uce_erase_embeds`), from the stored rows + names."""
    m = c.meta
    emb = {}
    for names, arr in ((m["edit"], c.arr("C_edit")), (m["debias"], c.arr("C_debias")), (m["preserve"], c.arr("C_pres"))):
        for n, r in zip(names, arr):
            emb[n] = torch.from_numpy(np.ascontiguousarray(r[None]))
    return emb
