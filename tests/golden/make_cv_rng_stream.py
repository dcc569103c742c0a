"""Generates cv_rng_mwc_stream.json: the first 64 outputs of cv::RNG for the seed that
cv::estimateAffinePartial2D constructs its RANSAC registrator with (`RNG rng((uint64)-1)`), from the
published recurrence  state = (uint32)state * 4164903690 + (state >> 32),  output = (uint32)state.
SURVEY.md 8a row R1 quotes the first four values (0x07c09cf5, 0xbac3439c, 0x99ae7b8c, 0x37275f45)."""
import json

s = (1 << 64) - 1
out = []
for _ in range(64):
    s = ((s & 0xFFFFFFFF) * 4164903690 + (s >> 32)) & ((1 << 64) - 1)
    out.append(s & 0xFFFFFFFF)
json.dump({"seed": "0xffffffffffffffff", "multiplier": 4164903690, "next_u32": out,
           "uniform_0_200": [x % 200 for x in out]}, open("cv_rng_mwc_stream.json", "w"), indent=0)
