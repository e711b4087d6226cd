// parakeet/parakeet.hpp -- umbrella header of the drop-in facade (reference: include/parakeet/parakeet.hpp).
#pragma once
#include "config.hpp"
#include "timestamp.hpp"
#include "transcribe.hpp"
#include "audio_io.hpp"
#include "audio.hpp"
#include "nemotron.hpp"
#include "sortformer.hpp"
#include "diarize.hpp"
#include "vocab.hpp"
