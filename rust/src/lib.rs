//! `orc_format::read::decode` iterators backed by the MI355X HIP decoder (liborcd.so).
//!
//! Every name the crate exports from that module (src/read/decode/mod.rs:8-14) with the same
//! constructor signatures -- BooleanIter, BooleanRleRunIter, BooleanRun, Float, FloatIter,
//! SignedRleV2Iter, SignedRleV2Run, SignedRleV2RunIter, UnsignedRleV2Iter, UnsignedRleV2Run,
//! UnsignedRleV2RunIter, Values -- so that `use orc_format::read::decode::*` can be pointed here.  Each iterator drains its reader (one
//! stream of one column, as Column::get_stream hands it out), makes ONE call into the C ABI, and
//! then iterates the decoded buffer.
//!
//! UNCOMPILED: this build environment has no Rust toolchain; the file documents the binding a
//! maintainer of the crate would add and has never been through rustc.
mod ffi;

use std::io::Read;
use std::os::raw::c_int;

#[derive(Debug, Clone)]
pub enum Error {
    OutOfSpec,
    InvalidUtf8,
    InvalidColumn(u32),
    InvalidKind(u32),
    DecodeFloat,
    Decompression,
    InvalidProto,
    Unsupported,
    Device,
}

impl Error {
    fn from_status(status: c_int) -> Self {
        match status {
            ffi::ORCD_OUT_OF_SPEC => Error::OutOfSpec,
            ffi::ORCD_INVALID_UTF8 => Error::InvalidUtf8,
            ffi::ORCD_INVALID_COLUMN => Error::InvalidColumn(0),
            ffi::ORCD_INVALID_KIND => Error::InvalidKind(0),
            ffi::ORCD_DECODE_FLOAT => Error::DecodeFloat,
            ffi::ORCD_DECOMPRESSION => Error::Decompression,
            ffi::ORCD_INVALID_PROTO => Error::InvalidProto,
            ffi::ORCD_UNSUPPORTED => Error::Unsupported,
            _ => Error::Device,
        }
    }
}

thread_local! {
    // one decode context per thread (orcd.h: a context belongs to one thread and one GPU)
    static CTX: *mut ffi::orcd_ctx = {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { ffi::orcd_ctx_create(0, &mut ctx) };
        assert_eq!(rc, ffi::ORCD_OK, "orcd_ctx_create failed");
        ctx
    };
}

fn ctx() -> *mut ffi::orcd_ctx {
    CTX.with(|c| *c)
}

struct Decoded<T> {
    values: std::vec::IntoIter<T>,
    status: c_int,
}

impl<T> Decoded<T> {
    fn next(&mut self) -> Option<Result<T, Error>> {
        if self.status != ffi::ORCD_OK {
            let e = Error::from_status(self.status);
            self.status = ffi::ORCD_OK;
            self.values = Vec::new().into_iter();
            return Some(Err(e));
        }
        self.values.next().map(Ok)
    }
}

fn decode_rlev2<R: Read>(reader: &mut R, length: usize, scratch: &mut Vec<u8>, signed: bool) -> Decoded<i64> {
    scratch.clear();
    let mut values = vec![0i64; length];
    let status = match reader.read_to_end(scratch) {
        Err(_) => ffi::ORCD_OUT_OF_SPEC,
        Ok(_) => unsafe {
            ffi::orcd_rlev2_decode(ctx(), scratch.as_ptr(), scratch.len(), length, std::ptr::null(), 0,
                                   values.as_mut_ptr(), if signed { ffi::ORCD_SIGNED } else { 0 })
        },
    };
    Decoded { values: values.into_iter(), status }
}

/// SignedRleV2Iter::new(reader, length, scratch) -- src/read/decode/rle_v2.rs:713-758
pub struct SignedRleV2Iter<R: Read> {
    reader: R,
    scratch: Vec<u8>,
    decoded: Decoded<i64>,
}

impl<R: Read> SignedRleV2Iter<R> {
    pub fn new(mut reader: R, length: usize, mut scratch: Vec<u8>) -> Self {
        let decoded = decode_rlev2(&mut reader, length, &mut scratch, true);
        Self { reader, scratch, decoded }
    }
    pub fn into_inner(mut self) -> (R, Vec<u8>) {
        self.scratch.clear();
        (self.reader, self.scratch)
    }
}

impl<R: Read> Iterator for SignedRleV2Iter<R> {
    type Item = Result<i64, Error>;
    fn next(&mut self) -> Option<Self::Item> {
        self.decoded.next()
    }
}

/// UnsignedRleV2Iter::new(reader, length, scratch) -- src/read/decode/rle_v2.rs:517-562
pub struct UnsignedRleV2Iter<R: Read> {
    reader: R,
    scratch: Vec<u8>,
    decoded: Decoded<i64>,
}

impl<R: Read> UnsignedRleV2Iter<R> {
    pub fn new(mut reader: R, length: usize, mut scratch: Vec<u8>) -> Self {
        let decoded = decode_rlev2(&mut reader, length, &mut scratch, false);
        Self { reader, scratch, decoded }
    }
    pub fn into_inner(mut self) -> (R, Vec<u8>) {
        self.scratch.clear();
        (self.reader, self.scratch)
    }
}

impl<R: Read> Iterator for UnsignedRleV2Iter<R> {
    type Item = Result<u64, Error>;
    fn next(&mut self) -> Option<Self::Item> {
        self.decoded.next().map(|r| r.map(|v| v as u64))
    }
}

/// BooleanIter::new(reader, length) -- src/read/decode/boolean_rle.rs:65-158
pub struct BooleanIter<R: Read> {
    reader: R,
    decoded: Decoded<u8>,
}

impl<R: Read> BooleanIter<R> {
    pub fn new(mut reader: R, length: usize) -> Self {
        let mut stream = Vec::new();
        let mut values = vec![0u8; length];
        let status = match reader.read_to_end(&mut stream) {
            Err(_) => ffi::ORCD_OUT_OF_SPEC,
            Ok(_) => unsafe {
                ffi::orcd_boolean_decode(ctx(), stream.as_ptr(), stream.len(), length, std::ptr::null(), 0,
                                         values.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut(), 0)
            },
        };
        Self { reader, decoded: Decoded { values: values.into_iter(), status } }
    }
    pub fn into_inner(self) -> R {
        self.reader
    }
}

impl<R: Read> Iterator for BooleanIter<R> {
    type Item = Result<bool, Error>;
    fn next(&mut self) -> Option<Self::Item> {
        self.decoded.next().map(|r| r.map(|b| b != 0))
    }
}

/// Sealed trait over f32 and f64 -- src/read/decode/float.rs:3-31
pub trait Float: Default + Copy + private::Sealed {
    type Bytes: AsRef<[u8]> + AsMut<[u8]> + Default;
    fn from_le_bytes(bytes: Self::Bytes) -> Self;
}

mod private {
    pub trait Sealed {}
    impl Sealed for f32 {}
    impl Sealed for f64 {}
}

impl Float for f32 {
    type Bytes = [u8; 4];
    #[inline]
    fn from_le_bytes(bytes: Self::Bytes) -> Self {
        Self::from_le_bytes(bytes)
    }
}

impl Float for f64 {
    type Bytes = [u8; 8];
    #[inline]
    fn from_le_bytes(bytes: Self::Bytes) -> Self {
        Self::from_le_bytes(bytes)
    }
}

/// FloatIter::<f32 | f64, R>::new(reader, length) -- src/read/decode/float.rs:34-91
pub struct FloatIter<T: Float, R: Read> {
    reader: R,
    decoded: Decoded<T>,
}

impl<T: Float, R: Read> FloatIter<T, R> {
    pub fn new(mut reader: R, length: usize) -> Self {
        let mut stream = Vec::new();
        let mut values = vec![T::default(); length];
        let status = match reader.read_to_end(&mut stream) {
            Err(_) => ffi::ORCD_DECODE_FLOAT,
            Ok(_) => unsafe {
                ffi::orcd_float_decode(ctx(), stream.as_ptr(), stream.len(), std::mem::size_of::<T>(), length,
                                       values.as_mut_ptr() as *mut _, 0)
            },
        };
        Self { reader, decoded: Decoded { values: values.into_iter(), status } }
    }
    /// The number of items remaining -- float.rs:51-55
    pub fn len(&self) -> usize {
        self.decoded.values.len()
    }
    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }
    pub fn into_inner(self) -> R {
        self.reader
    }
}

impl<T: Float, R: Read> Iterator for FloatIter<T, R> {
    type Item = Result<T, Error>;
    fn next(&mut self) -> Option<Self::Item> {
        self.decoded.next()
    }
    fn size_hint(&self) -> (usize, Option<usize>) {
        (self.len(), Some(self.len()))
    }
}

// ---- runs of an RLE v2 stream ---------------------------------------------------------------------

/// The values of one run, already decoded: what UnsignedDirectRun / UnsignedDeltaRun /
/// UnsignedShortRepeat and their signed twins iterate (src/read/decode/rle_v2.rs:137-175, 226-281,
/// 283-330, 565-627).  `len()` is the number of values left, as there.
#[derive(Debug, Clone)]
pub struct RunValues<T> {
    values: std::vec::IntoIter<T>,
}

impl<T> RunValues<T> {
    pub fn len(&self) -> usize {
        self.values.len()
    }
    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }
}

impl<T> Iterator for RunValues<T> {
    type Item = T;
    fn next(&mut self) -> Option<T> {
        self.values.next()
    }
    fn size_hint(&self) -> (usize, Option<usize>) {
        (self.len(), Some(self.len()))
    }
}

/// UnsignedRleV2Run -- src/read/decode/rle_v2.rs:432-439 (PatchedBase, which the crate leaves
/// `todo!()` at rle_v2.rs:455, is decoded here and gets a variant of its own).
#[derive(Debug, Clone)]
pub enum UnsignedRleV2Run {
    Direct(RunValues<u64>),
    Delta(RunValues<u64>),
    ShortRepeat(RunValues<u64>),
    PatchedBase(RunValues<u64>),
}

impl UnsignedRleV2Run {
    /// The number of items remaining -- rle_v2.rs:467-477
    pub fn len(&self) -> usize {
        match self {
            Self::Direct(r) | Self::Delta(r) | Self::ShortRepeat(r) | Self::PatchedBase(r) => r.len(),
        }
    }
    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }
}

/// SignedRleV2Run -- src/read/decode/rle_v2.rs:629-636
#[derive(Debug, Clone)]
pub enum SignedRleV2Run {
    Direct(RunValues<i64>),
    Delta(RunValues<i64>),
    ShortRepeat(RunValues<i64>),
    PatchedBase(RunValues<i64>),
}

impl SignedRleV2Run {
    /// The number of items remaining -- rle_v2.rs:664-674
    pub fn len(&self) -> usize {
        match self {
            Self::Direct(r) | Self::Delta(r) | Self::ShortRepeat(r) | Self::PatchedBase(r) => r.len(),
        }
    }
    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }
}

/// Values and run table of one stream: two calls into the C ABI (the decode, and the header scan
/// alone for the table), however many runs there are.
struct DecodedRuns {
    values: Vec<i64>,
    table: std::vec::IntoIter<ffi::orcd_run>,
    status: c_int,
}

fn decode_runs<R: Read>(reader: &mut R, length: usize, scratch: &mut Vec<u8>, signed: bool) -> DecodedRuns {
    scratch.clear();
    let flags = if signed { ffi::ORCD_SIGNED } else { 0 };
    let mut values = vec![0i64; length];
    let mut table: Vec<ffi::orcd_run> = Vec::new();
    let mut status = match reader.read_to_end(scratch) {
        Err(_) => ffi::ORCD_OUT_OF_SPEC,
        Ok(_) => unsafe {
            ffi::orcd_rlev2_decode(ctx(), scratch.as_ptr(), scratch.len(), length, std::ptr::null(), 0,
                                   values.as_mut_ptr(), flags)
        },
    };
    if status == ffi::ORCD_OK && length > 0 {
        // a run takes at least two stream bytes and holds at least one value
        let cap = std::cmp::min(length, scratch.len() / 2 + 1);
        table.resize(cap, ffi::orcd_run::default());
        let mut n_runs = 0usize;
        status = unsafe {
            ffi::orcd_rlev2_runs(ctx(), scratch.as_ptr(), scratch.len(), length, std::ptr::null(), 0,
                                 table.as_mut_ptr(), cap, &mut n_runs, flags)
        };
        table.truncate(std::cmp::min(n_runs, cap));
    }
    DecodedRuns { values, table: table.into_iter(), status }
}

impl DecodedRuns {
    /// The next run's kind and values; the last run is cut at `length` values (the crate's run
    /// iterators hand out whole runs and let the value iterator stop: rle_v2.rs:536-541).
    fn next(&mut self) -> Option<Result<(u32, Vec<i64>), Error>> {
        if self.status != ffi::ORCD_OK {
            let e = Error::from_status(self.status);
            self.status = ffi::ORCD_OK;
            self.table = Vec::new().into_iter();
            return Some(Err(e));
        }
        let run = self.table.next()?;
        let beg = run.out_offset as usize;
        let end = std::cmp::min(beg + run.length as usize, self.values.len());
        Some(Ok((run.kind, self.values[beg..end].to_vec())))
    }
}

/// UnsignedRleV2RunIter::new(reader, length, scratch) -- src/read/decode/rle_v2.rs:480-514
pub struct UnsignedRleV2RunIter<R: Read> {
    reader: R,
    scratch: Vec<u8>,
    decoded: DecodedRuns,
}

impl<R: Read> UnsignedRleV2RunIter<R> {
    pub fn new(mut reader: R, length: usize, mut scratch: Vec<u8>) -> Self {
        let decoded = decode_runs(&mut reader, length, &mut scratch, false);
        Self { reader, scratch, decoded }
    }
    pub fn into_inner(mut self) -> (R, Vec<u8>) {
        self.scratch.clear();
        (self.reader, self.scratch)
    }
}

impl<R: Read> Iterator for UnsignedRleV2RunIter<R> {
    type Item = Result<UnsignedRleV2Run, Error>;
    fn next(&mut self) -> Option<Self::Item> {
        self.decoded.next().map(|r| {
            r.map(|(kind, values)| {
                let values: Vec<u64> = values.into_iter().map(|v| v as u64).collect();
                let run = RunValues { values: values.into_iter() };
                match kind {
                    0 => UnsignedRleV2Run::ShortRepeat(run),
                    1 => UnsignedRleV2Run::Direct(run),
                    2 => UnsignedRleV2Run::PatchedBase(run),
                    _ => UnsignedRleV2Run::Delta(run),
                }
            })
        })
    }
}

/// SignedRleV2RunIter::new(reader, length, scratch) -- src/read/decode/rle_v2.rs:677-710
pub struct SignedRleV2RunIter<R: Read> {
    reader: R,
    scratch: Vec<u8>,
    decoded: DecodedRuns,
}

impl<R: Read> SignedRleV2RunIter<R> {
    pub fn new(mut reader: R, length: usize, mut scratch: Vec<u8>) -> Self {
        let decoded = decode_runs(&mut reader, length, &mut scratch, true);
        Self { reader, scratch, decoded }
    }
    pub fn into_inner(mut self) -> (R, Vec<u8>) {
        self.scratch.clear();
        (self.reader, self.scratch)
    }
}

impl<R: Read> Iterator for SignedRleV2RunIter<R> {
    type Item = Result<SignedRleV2Run, Error>;
    fn next(&mut self) -> Option<Self::Item> {
        self.decoded.next().map(|r| {
            r.map(|(kind, values)| {
                let run = RunValues { values: values.into_iter() };
                match kind {
                    0 => SignedRleV2Run::ShortRepeat(run),
                    1 => SignedRleV2Run::Direct(run),
                    2 => SignedRleV2Run::PatchedBase(run),
                    _ => SignedRleV2Run::Delta(run),
                }
            })
        })
    }
}

// ---- runs of a byte-RLE stream --------------------------------------------------------------------

/// BooleanRun -- src/read/decode/boolean_rle.rs:7-12.  `Literals` holds the literal bytes in front
/// and zeros behind them, as the crate's 255-byte array does.
#[derive(Debug, Copy, Clone, PartialEq)]
#[allow(clippy::large_enum_variant)]
pub enum BooleanRun {
    Run(u8, u16),
    Literals([u8; 255]),
}

/// BooleanRleRunIter::new(reader) -- src/read/decode/boolean_rle.rs:14-63.  One control-byte scan
/// on the GPU for the whole stream (`orcd_byte_rle_runs`); the run's bytes are then read from the
/// stream at the positions the table gives.  A last run that is cut short is `Err(OutOfSpec)`;
/// at the end of the stream the iterator ends (the crate's yields an Err from then on).
pub struct BooleanRleRunIter<R: Read> {
    reader: R,
    stream: Vec<u8>,
    table: std::vec::IntoIter<ffi::orcd_byte_run>,
    status: c_int,
}

impl<R: Read> BooleanRleRunIter<R> {
    pub fn new(mut reader: R) -> Self {
        let mut stream = Vec::new();
        let mut table: Vec<ffi::orcd_byte_run> = Vec::new();
        let status = match reader.read_to_end(&mut stream) {
            Err(_) => ffi::ORCD_OUT_OF_SPEC,
            Ok(_) => {
                let cap = stream.len() / 2 + 1; // a run takes at least two stream bytes
                table.resize(cap, ffi::orcd_byte_run::default());
                let mut n_runs = 0usize;
                let rc = unsafe {
                    ffi::orcd_byte_rle_runs(ctx(), stream.as_ptr(), stream.len(), ffi::ORCD_WHOLE_STREAM,
                                            std::ptr::null(), 0, table.as_mut_ptr(), cap, &mut n_runs,
                                            std::ptr::null_mut(), 0)
                };
                table.truncate(std::cmp::min(n_runs, cap));
                rc
            }
        };
        Self { reader, stream, table: table.into_iter(), status }
    }

    pub fn into_inner(self) -> R {
        self.reader
    }
}

impl<R: Read> Iterator for BooleanRleRunIter<R> {
    type Item = Result<BooleanRun, Error>;
    fn next(&mut self) -> Option<Self::Item> {
        if self.status != ffi::ORCD_OK {
            let e = Error::from_status(self.status);
            self.status = ffi::ORCD_OK;
            self.table = Vec::new().into_iter();
            return Some(Err(e));
        }
        let run = self.table.next()?;
        let at = run.payload_offset as usize;
        if run.is_run != 0 {
            Some(Ok(BooleanRun::Run(self.stream[at], run.length as u16)))
        } else {
            let mut literals = [0u8; 255];
            let n = run.length as usize;
            literals[..n].copy_from_slice(&self.stream[at..at + n]);
            Some(Ok(BooleanRun::Literals(literals)))
        }
    }
}

// ---- variable-length values -------------------------------------------------------------------------

/// Values::new(reader, scratch) -- src/read/decode/variable_length.rs:5-28.  Host bytes in, host
/// bytes out: there is nothing to decode, so this is the crate's own logic; whole string columns go
/// through `orcd_string_direct_decode` / `orcd_decode_column` instead of one `next` per value.
pub struct Values<R: Read> {
    reader: R,
    scratch: Vec<u8>,
}

impl<R: Read> Values<R> {
    pub fn new(reader: R, scratch: Vec<u8>) -> Self {
        Self { reader, scratch }
    }

    pub fn next(&mut self, length: usize) -> Result<&[u8], Error> {
        self.scratch.clear();
        self.scratch.reserve(length);
        (&mut self.reader)
            .take(length as u64)
            .read_to_end(&mut self.scratch)
            .map_err(|_| Error::OutOfSpec)?;
        Ok(&self.scratch)
    }

    pub fn into_inner(self) -> Vec<u8> {
        self.scratch
    }
}
