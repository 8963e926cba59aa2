'use strict';
// The geometry / validation / error contract that the reference's web tests expect of the absent
// `web/src/lib/videoProcessor` (web/src/tests/video/compression.test.js:31-126,
// portrait.test.js:31-156), over the C ABI, for decoded yuv420p frames.  Container, codec, audio
// and the File object stay with the host; what is kept is: which filter string a clip gets
// (portrait.test.js:70-97,133-147), the portrait checks, and the four error names.
const native = require('./index.js');

const VIDEO_CONSTANTS = {                      // web/src/tests/video/__mocks__/ffmpeg.js:94-108
  SUPPORTED_FORMATS: ['video/mp4', 'video/quicktime', 'video/x-m4v'],
  DEFAULT_DIMENSIONS: { maxWidth: 720, maxHeight: 1280 },
  ASPECT_RATIO: { PORTRAIT: 16 / 9, TOLERANCE: 0.1 },
  MAX_SIZE_BYTES: 100 * 1024 * 1024,           // compression.test.js:32,45: 150 MB is compressed, 50 MB is returned as it is
};

function fail(name, detail) { const e = new Error(`${name}: ${detail}`); e.code = name; return e; }

// metadata: { width, height } of the first video stream
function portraitInfo(metadata) {
  if (!metadata || !(metadata.width > 0) || !(metadata.height > 0)) throw new Error('Invalid video metadata');
  const ratio = metadata.height / metadata.width;
  const want = VIDEO_CONSTANTS.ASPECT_RATIO.PORTRAIT;
  return { isPortrait: metadata.height > metadata.width, ratio,
           isValid: Math.abs(ratio - want) / want <= VIDEO_CONSTANTS.ASPECT_RATIO.TOLERANCE };
}

// The spec's validatePortraitMode takes the file and rejects: 'Invalid video file' for a zero-byte one
// (portrait.test.js:127-131), 'Invalid video metadata' when the probe gives no size (:148-154).  A clip here
// carries its probe result (width, height) with it.
async function validatePortraitMode(clip) {
  if (!clip) throw new Error('Invalid video file');
  const bytes = Buffer.isBuffer(clip.data) ? clip.data.length : clip.size;
  if (bytes === 0) throw new Error('Invalid video file');
  return portraitInfo(clip);
}

// 9:16 sources are scaled to the box; anything else is fitted inside it (portrait.test.js:85-97,133-147)
function scaleFilter(metadata, dims) {
  const d = Object.assign({}, VIDEO_CONSTANTS.DEFAULT_DIMENSIONS, dims);
  if (portraitInfo(metadata).isValid) return `scale=${d.maxWidth}:${d.maxHeight}`;
  return `scale='min(${d.maxWidth},iw)':'min(${d.maxHeight},ih)':force_original_aspect_ratio=decrease`;
}

// clip: { type, width, height, data: Buffer of packed yuv420p frames }.  Resolves { width, height, data, vf }.
async function compress(clip, options) {
  options = options || {};
  if (!clip || !Buffer.isBuffer(clip.data)) throw fail('INVALID_INPUT', 'clip with a data Buffer expected');
  if (!VIDEO_CONSTANTS.SUPPORTED_FORMATS.includes(clip.type)) throw fail('INVALID_FORMAT', `unsupported type ${clip.type}`);
  const frame = native.frameBytes(clip.width | 0, clip.height | 0);
  if (!frame || clip.data.length === 0 || clip.data.length % frame) throw fail('INVALID_INPUT', 'data is not a whole number of frames');
  // Size gate (compression.test.js:44-50): a file at or under the limit comes back untouched -- the same object,
  // nothing run.  `size` is the container's byte size where the host knows it, else the raw frames'.
  const maxSize = options.maxSizeBytes === undefined ? VIDEO_CONSTANTS.MAX_SIZE_BYTES : options.maxSizeBytes;
  const size = typeof clip.size === 'number' ? clip.size : clip.data.length;
  if (size <= maxSize) return clip;
  const vf = scaleFilter(clip, options);
  if (options.onProgress) options.onProgress(0);
  let geom, data;
  try {
    geom = native.parseVf(vf, clip.width, clip.height);
    data = native.runHost(vf, clip.width, clip.height, clip.data);
  } catch (e) {                                          // the library's status names are the spec's own
    const m = /^(INVALID_INPUT|INVALID_FORMAT|INITIALIZATION_FAILED|PROCESSING_FAILED)/.exec(e.message);
    throw fail(m ? m[1] : 'PROCESSING_FAILED', e.message);
  }
  if (options.onProgress) options.onProgress(100);
  return { type: 'video/mp4', width: geom.out_w, height: geom.out_h, data, vf };
}

// terminate(): compression.test.js:27-29 calls it after every test; drops the cached plans and the staging pool
module.exports = { videoProcessor: { compress, validatePortraitMode, scaleFilter, terminate() { return native.terminate(); } }, VIDEO_CONSTANTS };
