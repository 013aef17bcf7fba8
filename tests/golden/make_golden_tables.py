#!/usr/bin/env python
"""The fixed-strength evaluation tables of the UNMODIFIED reference (videoseal/augmentation/__init__.py:12-130) as class names + parameters,
for tests/test_host.py::test_validation_tables_match_the_reference (needs /root/reference):

    python tests/golden/make_golden_tables.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                                   # noqa: E402
import make_golden_fwd as MF                               # noqa: E402


def rows(table):
    out = []
    for aug, params in table:
        if aug.__class__.__name__ == "Sequential":
            name = "Sequential(" + ",".join(t.__class__.__name__ for t in aug.transforms) + ")"
        else:
            name = aug.__class__.__name__
        out.append([name, [list(p) if isinstance(p, tuple) else p for p in params]])
    return out


def main():
    MG.import_reference()
    MF.patch_torchvision()
    import videoseal.augmentation as A
    d = {"validation_image": rows(A.get_validation_augs(False)), "validation_video": rows(A.get_validation_augs(True)),
         "identity": rows(A.get_validation_augs(False, only_identity=True)),
         "combined_image": rows(A.get_validation_augs(False, only_combined=True)), "combined_video": rows(A.get_validation_augs(True, only_combined=True)),
         "subset_image": rows(A.get_validation_augs_subset(False)), "subset_video": rows(A.get_validation_augs_subset(True))}
    with open(os.path.join(HERE, "validation_tables.json"), "w") as f:
        json.dump(d, f, indent=1)
    print({k: len(v) for k, v in d.items()})


if __name__ == "__main__":
    main()
