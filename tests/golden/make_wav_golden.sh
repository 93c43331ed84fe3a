#!/bin/sh
# Writes tests/golden/audiofile_24bit_v1.wav: the file the REFERENCE's AudioFile (AudioFile.cpp / AudioFile.h, compiled where they
# lie and as they stand - build container only, /root/reference does not travel) writes for tests/golden/wav_samples.inc through
# the calls of the reference's outputAudioFile (main.cpp:337-366).
set -e
here=$(cd "$(dirname "$0")" && pwd)
ref=${SOTS_REFERENCE:-/root/reference}
tmp=$(mktemp -d)
g++ -std=c++17 -O1 -DSOTS_WAV_REFERENCE -I"$ref" -I"$here" -o "$tmp/wavref" "$here/wav_driver.cpp" "$ref/AudioFile.cpp"
"$tmp/wavref" "$here/audiofile_24bit_v1.wav"
rm -rf "$tmp"
ls -l "$here/audiofile_24bit_v1.wav"
