"""Public API of fish-tts on the MI355X path: get_instance / reset_instance / FishTTS / VoiceProfile with the
signatures, defaults and exceptions of fish_tts/synthesizer.py:47-719.  The dual-AR decode and the DAC decode
run in libfishtts_hip.so; this file is host logic only (threads, queues, WAV packing)."""
from __future__ import annotations

import io
import logging
import queue
import threading
import time
import wave
from dataclasses import dataclass, field
from pathlib import Path
from typing import Iterator, List, Literal, Optional

import numpy as np

logger = logging.getLogger(__name__)

_instance: "FishTTS | None" = None
_instance_lock = threading.Lock()


@dataclass
class VoiceProfile:
    """Encoded reference audio: (num_codebooks, T) integer codes + transcript (synthesizer.py:47-65)."""
    codes: np.ndarray
    text: str = ""
    name: str = ""

    def save(self, path) -> None:
        np.save(path, self.codes)

    @classmethod
    def load(cls, path, text: str = "", name: str = "") -> "VoiceProfile":
        codes = np.load(path)
        if not name:
            name = Path(path).stem
        return cls(codes=codes, text=text, name=name)


@dataclass
class _PrefillCache:
    prompt_text: List[str] = field(default_factory=list)
    prompt_tokens: List[np.ndarray] = field(default_factory=list)
    profiles: List[VoiceProfile] = field(default_factory=list)


class FishTTS:
    """TTS synthesizer: DualARTransformer + DAC vocoder on one MI355X (one process per GPU)."""

    def __init__(self, model_dir=None, device: Literal["cpu", "cuda"] = "cuda",
                 precision: Literal["bf16", "fp16", "fp32"] = "bf16", warmup: bool = True, *,
                 _synthetic: Optional[dict] = None, gpu_index: int = 0, cache_reference_kv: bool = True,
                 max_batch: int = 1, batch_streams: int = 1):
        """`max_batch` (extension): utterance slots for synthesize_batch (lock-step batch with refill).
        `batch_streams` (extension): synthesize_batch of more than `max_batch` texts runs that many lock-step batches
        side by side (one engine - context, stream, weight copy - per batch, created on first use): a lock-step frame
        leaves most of the chip idle, so independent batches overlap (batch.run_batch_streams; three is the measured
        optimum at 32 slots each).
        `cache_reference_kv` (extension): keep the K/V of the reference part of the prompt on
        the device per voice, so a cloned-voice call prefills only the new text (the reference re-prefills ~700
        prompt positions per call: synthesizer.py:363-377, inference.py:779-793)."""
        from .generation import PrefixCache
        self._prefix_cache = PrefixCache() if cache_reference_kv else None
        self._max_batch = int(max_batch)
        self._batch_streams = max(1, int(batch_streams))
        self._more_engines: list = []
        self._engine_factory = None
        self.device = device
        self._precision = precision
        self._warmup = warmup
        self._engine = None
        self._tokenizer = None
        self._vocoder = None
        self._is_warmed_up = False
        self._prefill_cache = _PrefillCache()
        self._prefill_lock = threading.Lock()
        self._gen_lock = threading.Lock()  # AR calls on one context are serialised
        self._gpu_index = gpu_index
        if device != "cuda":
            raise RuntimeError("fish_tts_amd runs the hot path on an MI355X only: device must be 'cuda' "
                               "(there is no CPU fallback)")
        if _synthetic is not None:
            self._load_synthetic(**_synthetic)
        else:
            self._model_dir = self._ensure_model(model_dir)
            self._load_models()
        if warmup:
            self._run_warmup()

    # ------------------------------------------------------------------ loading
    def _ensure_model(self, model_dir) -> Path:
        if model_dir is not None:
            return Path(model_dir)
        # the reference downloads fishaudio/openaudio-s1-mini here (synthesizer.py:146-156)
        raise RuntimeError("model_dir is required: this build does not download checkpoints "
                           "(pass the directory holding config.json / model.pth / codec.pth / tokenizer.tiktoken)")

    def _load_models(self) -> None:
        import torch

        from .ar_engine import ARHipEngine
        from .codec_engine import CodecHipEngine
        from .config import DualARModelArgs
        from .tokenizer import IM_END_TOKEN, load_tokenizer
        from .weights import load_checkpoint
        t0 = time.perf_counter()
        args = DualARModelArgs.from_pretrained(str(self._model_dir))
        self._tokenizer = load_tokenizer(self._model_dir)
        tok = self._tokenizer

        def make_engine():
            eng = ARHipEngine(args, tok.semantic_begin_id, tok.semantic_end_id, tok.get_token_id(IM_END_TOKEN),
                              precision=self._precision, device=self._gpu_index, max_batch=self._max_batch,
                              max_new_tokens=2048 + 8)
            eng.load_state_dict(load_checkpoint(self._model_dir))
            return eng
        self._engine_factory = make_engine
        self._engine = make_engine()
        logger.info("Transformer loaded in %.1fs", time.perf_counter() - t0)
        codec_path = self._model_dir / "codec.pth"
        if codec_path.exists():
            sd = torch.load(codec_path, map_location="cpu", weights_only=True)
            inner = sd["state_dict"] if "state_dict" in sd else sd
            has_encoder = any(k.startswith(("encoder.", "generator.encoder.")) for k in inner)
            self._vocoder = CodecHipEngine(device=self._gpu_index, max_frames=2048 + 8, with_encoder=has_encoder)
            self._vocoder.load_state_dict(sd)
            logger.info("Vocoder loaded (bf16 contractions, f32 accumulate)")
        else:
            logger.warning("codec.pth not found, vocoder not loaded")

    def _load_synthetic(self, args, tokenizer, codec_args=None, seed: int = 0, with_codec: bool = True,
                        max_new_tokens: int = 2048 + 8, std=None, with_encoder: bool = False):
        from .ar_engine import ARHipEngine
        from .codec_engine import CodecHipEngine
        from .tokenizer import IM_END_TOKEN
        from .weights import random_state_dict
        import torch
        self._tokenizer = tokenizer
        dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(self._precision, torch.float32)

        def make_engine():
            eng = ARHipEngine(args, tokenizer.semantic_begin_id, tokenizer.semantic_end_id,
                              tokenizer.get_token_id(IM_END_TOKEN), precision=self._precision,
                              device=self._gpu_index, max_batch=self._max_batch, max_new_tokens=max_new_tokens)
            eng.load_state_dict(random_state_dict(args, seed=seed, dtype=dtype, std=std))
            return eng
        self._engine_factory = make_engine
        self._engine = make_engine()
        if with_codec:
            self._vocoder = CodecHipEngine.synthetic(device=self._gpu_index, max_frames=max_new_tokens, seed=seed,
                                                     args=codec_args, with_encoder=with_encoder)

    @classmethod
    def synthetic(cls, args, tokenizer, codec_args=None, precision="bf16", seed: int = 0, warmup: bool = False,
                  with_codec: bool = True, max_new_tokens: int = 2048 + 8, std=None, gpu_index: int = 0,
                  cache_reference_kv: bool = True, max_batch: int = 1, batch_streams: int = 1) -> "FishTTS":
        """Random-init model of the given shapes (no checkpoint on disk): benches, smoke tests."""
        return cls(None, "cuda", precision, warmup, gpu_index=gpu_index, cache_reference_kv=cache_reference_kv,
                   max_batch=max_batch, batch_streams=batch_streams,
                   _synthetic=dict(args=args, tokenizer=tokenizer, codec_args=codec_args, seed=seed,
                                   with_codec=with_codec, max_new_tokens=max_new_tokens, std=std))

    def _run_warmup(self) -> None:
        logger.info("Running warmup (captures the frame graph)...")
        t0 = time.perf_counter()
        try:
            from .generation import generate_long
            with self._gen_lock:
                for response in generate_long(engine=self._engine, tokenizer=self._tokenizer, text="Hello.",
                                              max_new_tokens=50, temperature=0.7, top_p=0.8, repetition_penalty=1.1,
                                              prompt_text=[], prompt_tokens=[]):
                    if response.action == "next":
                        break
            self._is_warmed_up = True
            logger.info("Warmup complete in %.1fs", time.perf_counter() - t0)
        except Exception as e:  # noqa: BLE001  (the reference only logs a failed warmup, synthesizer.py:322-323)
            logger.warning("Warmup failed: %s", e)

    def encode_reference(self, audio_bytes: bytes, text: str) -> VoiceProfile:
        """WAV bytes + transcript -> VoiceProfile (synthesizer.py:325-357): the codec *encoder* on the GPU
        (SURVEY.md §8-f F4)."""
        if self._vocoder is None:
            raise RuntimeError("Vocoder not loaded")
        audio = self._read_wav(audio_bytes)
        codes = self._vocoder.encode(audio)
        return VoiceProfile(codes=codes.astype(np.int64), text=text)

    @staticmethod
    def _read_wav(audio_bytes: bytes) -> np.ndarray:
        """synthesizer.py:613-631: 16-bit PCM -> float32 / 32768, Fourier resampling to 44.1 kHz if needed."""
        with wave.open(io.BytesIO(audio_bytes), "rb") as wf:
            sample_rate = wf.getframerate()
            audio = np.frombuffer(wf.readframes(wf.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0
        if sample_rate != 44100:
            from scipy import signal
            audio = signal.resample(audio, int(len(audio) * 44100 / sample_rate))
        return audio

    # ------------------------------------------------------------------ references (synthesizer.py:363-429)
    def set_references(self, profiles: List[VoiceProfile]) -> None:
        with self._prefill_lock:
            self._prefill_cache = _PrefillCache(prompt_text=[p.text for p in profiles],
                                                prompt_tokens=[np.asarray(p.codes) for p in profiles],
                                                profiles=list(profiles))
            logger.info("Set %d reference(s)", len(profiles))

    def add_reference(self, profile: VoiceProfile) -> None:
        with self._prefill_lock:
            self._prefill_cache.profiles.append(profile)
            self._prefill_cache.prompt_text.append(profile.text)
            self._prefill_cache.prompt_tokens.append(np.asarray(profile.codes))

    def clear_references(self) -> None:
        with self._prefill_lock:
            self._prefill_cache = _PrefillCache()

    def get_references(self) -> List[VoiceProfile]:
        with self._prefill_lock:
            return list(self._prefill_cache.profiles)

    @property
    def num_references(self) -> int:
        return len(self._prefill_cache.profiles)

    def _get_prompt_data(self, references):
        if references is not None:
            return [p.text for p in references], [np.asarray(p.codes) for p in references]
        with self._prefill_lock:
            return list(self._prefill_cache.prompt_text), list(self._prefill_cache.prompt_tokens)

    # ------------------------------------------------------------------ synthesis
    def synthesize(self, text: str, references: Optional[List[VoiceProfile]] = None, temperature: float = 0.7,
                   top_p: float = 0.8, repetition_penalty: float = 1.1, max_tokens: int = 2048) -> bytes:
        """Text -> WAV bytes (synthesizer.py:431-481)."""
        from .generation import generate_long
        prompt_text, prompt_tokens = self._get_prompt_data(references)
        codes_list = []
        with self._gen_lock:
            for response in generate_long(engine=self._engine, tokenizer=self._tokenizer, text=text,
                                          max_new_tokens=max_tokens, temperature=temperature, top_p=top_p,
                                          repetition_penalty=repetition_penalty, prompt_text=prompt_text,
                                          prompt_tokens=prompt_tokens, prefix_cache=self._prefix_cache):
                if response.action == "sample":
                    codes_list.append(response.codes)
                elif response.action == "next":
                    break
        if not codes_list:
            raise RuntimeError("No audio generated")
        return self._decode_to_wav(np.concatenate(codes_list, axis=1))

    def synthesize_batch(self, texts: List[str], references: Optional[List[VoiceProfile]] = None,
                         temperature: float = 0.7, top_p: float = 0.8, repetition_penalty: float = 1.1,
                         max_tokens: int = 2048, seed: int = 0, seeds: Optional[List[int]] = None) -> List[bytes]:
        """Extension (BASELINE configs[2]): many texts -> WAV bytes each, decoded `max_batch` at a time in lock step
        with refill (fish_tts_amd.batch); utterance i uses seed + i, or seeds[i] when `seeds` is given (a sharded run
        passes the GLOBAL indices so an utterance draws the same noise on any number of GPUs).  Same per-utterance
        semantics as synthesize()."""
        from .batch import Utterance, run_batch, run_batch_streams
        from .prompt import build_prompt_split
        assert 0 < top_p <= 1, "top_p must be in (0, 1]"
        assert 0 < repetition_penalty < 2, "repetition_penalty must be in (0, 2)"
        assert 0 < temperature < 2, "temperature must be in (0, 2)"
        if seeds is not None and len(seeds) != len(texts):
            raise ValueError("seeds must have one entry per text")
        prompt_text, prompt_tokens = self._get_prompt_data(references)
        ncb = self._engine.args.num_codebooks
        with self._gen_lock:
            utts = []
            engines = [self._engine]
            if self._batch_streams > 1 and len(texts) > self._max_batch:
                while len(self._more_engines) < self._batch_streams - 1:      # created on first use, kept
                    self._more_engines.append(self._engine_factory())
                engines += self._more_engines
            for i, text in enumerate(texts):
                enc, n_prefix = build_prompt_split(self._tokenizer, text, prompt_text, prompt_tokens, ncb)
                if enc.shape[1] > self._engine.args.max_seq_len - 2048:
                    raise ValueError(f"Prompt is too long: {enc.shape[1]} > {self._engine.args.max_seq_len - 2048}")
                prefix = None
                if self._prefix_cache is not None and n_prefix >= self._prefix_cache.min_positions:
                    # a saved prefix lives in one engine's memory and pins its utterance there: spread them evenly
                    prefix = self._prefix_cache.get(engines[i % len(engines)], enc[:, :n_prefix])
                utts.append(Utterance(enc, max_tokens, temperature, top_p, repetition_penalty, seeds[i] if seeds is not None else seed + i,
                                      prefix=prefix))
            if len(engines) > 1:
                run_batch_streams(engines, utts)
            else:
                run_batch(self._engine, utts)
        out = []
        for u in utts:
            codes = u.codes()
            if codes.shape[1] == 0:
                raise RuntimeError("No audio generated")
            out.append(self._decode_to_wav(codes))
        return out

    def synthesize_stream(self, text: str, references: Optional[List[VoiceProfile]] = None, chunk_tokens: int = 20,
                          min_first_chunk: int = 10, **kwargs) -> Iterator[bytes]:
        """Streaming synthesis: AR generation on this thread, codec decode on a worker thread with two bounded
        queues; each chunk is decoded independently from zero context (synthesizer.py:483-584).

        Extension `seamless=True` (keyword): the codec is strictly causal, so a chunk can be decoded with the context its
        predecessors left (the last 127 frames' K/V of the transformer layers, the last rows of every convolution input:
        CodecStream / ft_codec_stream_*) - no restart artefacts at chunk boundaries (the stateful streaming decode of
        SURVEY.md section 8-f F4), at the cost of one chunk each.  The result does not depend on the chunking, bit for
        bit; it equals the non-streaming decode of the same codes up to summation order (the stream always runs the
        kernel variants of a nominal 215-frame utterance: bit-equal at about that length, relative RMS <= 2e-2 measured at
        60 and 300 frames).  One stream carries at most `max_frames` frames (2056 here: the codec's rotation table); a
        longer synthesis starts a fresh stream there - that one boundary is decoded from zero state, as the reference
        decodes every chunk."""
        from .generation import generate_long
        seamless = bool(kwargs.get("seamless", False))
        prompt_text, prompt_tokens = self._get_prompt_data(references)
        codes_queue: "queue.Queue" = queue.Queue(maxsize=3)
        audio_queue: "queue.Queue" = queue.Queue(maxsize=3)
        error_holder: List[Exception] = []

        def decoder_worker():
            stream = None
            try:
                if seamless:
                    if self._vocoder is None:
                        raise RuntimeError("Vocoder not loaded")
                    stream = self._vocoder.stream()     # carried state: K/V of the last 127 frames, conv tails
                while True:
                    codes = codes_queue.get()
                    if codes is None:
                        break
                    if stream is None:
                        audio_queue.put(self._decode_to_pcm(codes))
                    else:
                        codes = np.asarray(codes)
                        if stream.frames + codes.shape[1] > self._vocoder.max_frames:   # the rotation table ends here
                            stream.close()
                            stream = self._vocoder.stream()
                        audio_queue.put((stream.decode(codes) * 32767).astype(np.int16).tobytes())
            except Exception as e:  # noqa: BLE001
                error_holder.append(e)
            finally:
                if stream is not None:
                    stream.close()
                audio_queue.put(None)

        worker = threading.Thread(target=decoder_worker, daemon=True)
        worker.start()

        def worker_gone() -> bool:
            # the worker records its exception BEFORE its final put(None) (which may itself wait for queue space)
            return bool(error_holder) or not worker.is_alive()

        pending: List[Optional[bytes]] = []   # audio taken off the queue while a put was waiting, in order

        def hand_over(item) -> bool:
            """Puts item on the bounded codes queue without ever blocking on a stuck or dead worker: keeps draining
            the audio queue meanwhile.  False = the worker is gone (its error is raised at the end)."""
            while True:
                if worker_gone():
                    return False
                try:
                    codes_queue.put(item, timeout=0.005)
                    return True
                except queue.Full:
                    pass
                try:
                    pending.append(audio_queue.get_nowait())
                except queue.Empty:
                    pass

        try:
            buffer, is_first_chunk, total_tokens = [], True, 0
            tail_chunk, worker_ok = None, True
            with self._gen_lock:
                for response in generate_long(engine=self._engine, tokenizer=self._tokenizer, text=text,
                                              max_new_tokens=kwargs.get("max_tokens", 2048),
                                              temperature=kwargs.get("temperature", 0.7), top_p=kwargs.get("top_p", 0.8),
                                              repetition_penalty=kwargs.get("repetition_penalty", 1.1),
                                              prompt_text=prompt_text, prompt_tokens=prompt_tokens, streaming=True,
                                              prefix_cache=self._prefix_cache):
                    if response.action == "sample":
                        buffer.append(response.codes)
                        total_tokens += response.codes.shape[1]
                        threshold = min_first_chunk if is_first_chunk else chunk_tokens
                        if total_tokens >= threshold:
                            chunk = np.concatenate(buffer, axis=1)
                            buffer, total_tokens, is_first_chunk = [], 0, False
                            # Both queues are bounded.  The reference blocks in put() here and joins the worker before
                            # draining (synthesizer.py:556-578): harmless when the AR loop is the slow side, a deadlock
                            # once it is faster than the codec (worker stuck on a full audio queue).  Keep draining; and
                            # stop generating when the worker has died (its exception is re-raised below, as the
                            # reference does at synthesizer.py:583-584).
                            worker_ok = hand_over(chunk)
                            while pending:
                                audio = pending.pop(0)
                                if audio is not None:
                                    yield audio
                            if not worker_ok:
                                break
                            while not audio_queue.empty():
                                audio = audio_queue.get_nowait()
                                if audio is not None:
                                    yield audio
                    elif response.action == "next":
                        if buffer:
                            tail_chunk = np.concatenate(buffer, axis=1)
                        break
        finally:
            # hand over the tail and the stop mark (the consumer may have abandoned the generator: then pending audio
            # is dropped); a dead worker takes nothing more
            for item in ([tail_chunk] if tail_chunk is not None else []) + [None]:
                if not hand_over(item):
                    break
        got_end = False
        for audio in pending:                 # drained while handing over, in order
            if audio is None:
                got_end = True
            else:
                yield audio
        while not got_end:                    # the worker always ends with a None
            try:
                audio = audio_queue.get(timeout=0.05)
            except queue.Empty:
                if worker.is_alive():
                    continue
                # the worker may have put its last chunk and the end mark between the time-out and this check
                while True:
                    try:
                        audio = audio_queue.get_nowait()
                    except queue.Empty:
                        break
                    if audio is not None:
                        yield audio
                break
            if audio is None:
                break
            yield audio
        worker.join()
        if error_holder:
            raise error_holder[0]

    # ------------------------------------------------------------------ codes -> audio (synthesizer.py:586-648)
    def _decode_to_wav(self, codes: np.ndarray) -> bytes:
        return self._to_wav_bytes(self._decode_codes(codes))

    def _decode_to_pcm(self, codes: np.ndarray) -> bytes:
        audio = self._decode_codes(codes)
        return (audio * 32767).astype(np.int16).tobytes()  # no clip on the PCM path (synthesizer.py:594)

    def _decode_codes(self, codes: np.ndarray) -> np.ndarray:
        if self._vocoder is None:
            raise RuntimeError("Vocoder not loaded")
        codes = np.asarray(codes)
        if codes.ndim == 2:
            codes = codes[None]
        return np.squeeze(self._vocoder.decode(codes))

    @staticmethod
    def _to_wav_bytes(audio: np.ndarray, sample_rate: int = 44100) -> bytes:
        audio = np.clip(audio, -1.0, 1.0)
        audio_int16 = (audio * 32767).astype(np.int16)
        buffer = io.BytesIO()
        with wave.open(buffer, "wb") as wf:
            wf.setnchannels(1)
            wf.setsampwidth(2)
            wf.setframerate(sample_rate)
            wf.writeframes(audio_int16.tobytes())
        return buffer.getvalue()

    @property
    def sample_rate(self) -> int:
        return 44100

    @property
    def precision(self) -> str:
        return self._precision


def get_instance(model_dir=None, device: Literal["cpu", "cuda"] = "cuda",
                 precision: Literal["bf16", "fp16", "fp32"] = "bf16", warmup: bool = True) -> FishTTS:
    """Process-wide singleton; later calls return the first instance and ignore their arguments
    (synthesizer.py:661-710)."""
    global _instance
    if _instance is not None:
        return _instance
    with _instance_lock:
        if _instance is not None:
            return _instance
        logger.info("Creating singleton FishTTS instance...")
        _instance = FishTTS(model_dir=model_dir, device=device, precision=precision, warmup=warmup)
        return _instance


def reset_instance() -> None:
    global _instance
    with _instance_lock:
        if _instance is not None:
            logger.info("Resetting singleton FishTTS instance")
            _instance = None
